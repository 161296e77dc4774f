"""Ad-hoc: print per-pass, per-surface difference stats GPU vs oracle (byte-level)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import parity as P
import test_gpu_parity as T
from oracle import okj_py as oracle
from kajiya_amd import lib as gpu
from kajiya_amd.abi import KJ_RTDGI_PASS
W, H = 256, 256
device = gpu.Device(0)
desc = T._scenes()["cornell"]
op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
fcs = T._frame_constants(W, H, 8)
repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
for fi, fc in enumerate(fcs):
    op.render_inputs(fc); op.reprojection(fc)
    gp.dev.frame_begin(fc)
    T._sync_inputs(op, gp, torch)
    repro_dev.copy_(torch.from_numpy(op.reprojection_map))
    gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
    if fi < 6:
        op.rtdgi_frame(fc); gp.rtdgi_frame(); torch.cuda.synchronize()
        T._upload_state(gp, T._oracle_surfaces(op), torch)
        continue
    T._upload_state(gp, T._oracle_surfaces(op), torch)
    op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
    gpu.check(gp.L.kj_rtdgi_reproject(gp.rtdgi, gp.reprojection_map_ptr, W, H, None))
    first = True
    for pname in T.PASS_ORDER:
        mask = KJ_RTDGI_PASS[pname] | (0 if first else T.KEEP); first = False
        before = T._oracle_surfaces(op)
        T._upload_state(gp, before, torch)
        p = op.params(mask); op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
        gpp = gp.params(mask); gpu.check(gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(gpp), C.byref(gp.out), None))
        torch.cuda.synchronize()
        ref = T._oracle_surfaces(op); got = T._download_state(gp, ref.keys(), torch)
        for n in ref:
            changed = (before[n] != ref[n]).sum()
            if changed == 0: continue
            r = P.compare(got[n], ref[n], P.fmt_of(n))
            print(f"frame {fi} {pname:>20s} {n:<36s} oracle-changed-bytes={changed:8d} gpu-vs-oracle differing bytes={(got[n]!=ref[n]).sum():7d} rel_l2={r['rel_l2']:.2e} mism={r['mismatch_frac']:.2e}")
