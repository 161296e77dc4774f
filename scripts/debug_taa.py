import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import parity as P, test_gpu_parity as T, test_gpu_taa as TT
from oracle import okj_py as oracle
from kajiya_amd import lib as gpu
W, H = 192, 160
device = gpu.Device(0)
op, gp = T._make_pipelines(gpu, oracle, device, T._scenes()["cornell"], W, H)
fcs = T._frame_constants(W, H, 4)
repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda"); inp_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
for fi, fc in enumerate(fcs):
    op.frame(fc); op.taa_frame(fc)
    gp.dev.frame_begin(fc)
    gp.depth.copy_(torch.from_numpy(op.depth)); repro_dev.copy_(torch.from_numpy(op.reprojection_map))
    gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
    inp_dev.copy_(torch.from_numpy(op.surface("spatial_filtered_tex", np.int16, (H, W, 4))))
    gp.taa_frame(input_ptr=inp_dev.data_ptr()); torch.cuda.synchronize()
    for name, fmt in TT.TAA_SURFACES.items():
        ref = op.taa_surface(name, np.uint8, (-1,)); got = gp.taa_surface(name, torch.uint8, (-1,)).cpu().numpy()
        if fmt == "r16f":
            a, b = TT._decode_r16f(got).astype(np.float64), TT._decode_r16f(ref).astype(np.float64)
        else:
            a, b = P.decode(got, fmt).astype(np.float64), P.decode(ref, fmt).astype(np.float64)
        d = np.abs(a - b); fin = np.isfinite(d)
        rel = np.sqrt((np.where(fin, d, 0) ** 2).sum()) / max(1e-12, np.sqrt((np.where(fin, b, 0) ** 2).sum()))
        nan_a, nan_b = np.isnan(a).sum(), np.isnan(b).sum()
        print(f"f{fi} {name:<30s} rel={rel:.2e} maxabs={np.nanmax(d):.3e} nan gpu/ref={nan_a}/{nan_b} bytes_diff={(got!=ref).sum()}")
        if name == "filtered_history_img" and fi == 1:
            idx = np.argsort(-np.where(fin, d, 0).max(axis=1))[:6]
            for i in idx: print("   px", i % W, i // W, "gpu", a[i], "ref", b[i])
            rh_a = P.decode(gp.taa_surface("reprojected_history_img", torch.uint8, (-1,)).cpu().numpy(), "rgba16f"); rh_b = P.decode(op.taa_surface("reprojected_history_img", np.uint8, (-1,)), "rgba16f")
            for i in idx[:3]:
                x, y = i % W, i // W
                for oy in (-1, 0, 1):
                    for ox in (-1, 0, 1):
                        j = (y + oy) * W + (x + ox)
                        if 0 <= j < W * H: print("      nb", ox, oy, "gpu", rh_a[j], "ref", rh_b[j])
