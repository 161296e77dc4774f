#!/usr/bin/env python3
"""Generate committed fixtures from the reference's DATA assets (run in the build
container only; /root/reference does not exist on the GPU box).

  kajiya_amd/data/bluenoise_256_rgba8.bin  <- assets/images/bluenoise/256_256/LDR_RGBA_0.png
        (bindless texture #1, default_world_renderer.rs:27; raw RGBA8, 262144 bytes)
  kajiya_amd/data/cornell_box.npz          <- assets/meshes/cornell_box/scene.{gltf,bin}
        imported the way kajiya-asset does (mesh.rs:279-437: node transforms baked,
        per-primitive material, winding flip on negative determinant), scale 2
        (assets/scenes/cornell_box.ron).
  kajiya_amd/data/pica_diorama.npz         <- assets/meshes/pica_pica_-_mini_diorama_01/scene.{gltf,bin}
        same import, scale 0.1 (assets/scenes/pica.ron). Geometry + per-material factors only: the six image maps
        (decals, one metallic-roughness map) are not carried, their materials keep the constant factors.
"""
import json, os, struct, sys
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kajiya_amd", "data")


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def node_matrix(n):
    if "matrix" in n:
        return np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T
    m = np.eye(4)
    s = np.array(n.get("scale", [1, 1, 1]), dtype=np.float64)
    r = quat_to_mat(n.get("rotation", [0, 0, 0, 1]))
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = n.get("translation", [0, 0, 0])
    return m


def load_gltf(path, scale):
    g = json.load(open(path))
    base = os.path.dirname(path)
    bufs = [open(os.path.join(base, b["uri"]), "rb").read() for b in g["buffers"]]

    def accessor(i):
        a = g["accessors"][i]
        bv = g["bufferViews"][a["bufferView"]]
        ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[a["type"]]
        dt = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}[a["componentType"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0)
        item = np.dtype(dt).itemsize * ncomp
        if stride and stride != item:
            raw = np.frombuffer(bufs[bv["buffer"]], dtype=np.uint8, offset=off, count=stride * (a["count"] - 1) + item)
            arr = np.stack([np.frombuffer(raw[k * stride:k * stride + item].tobytes(), dtype=dt) for k in range(a["count"])])
        else:
            arr = np.frombuffer(bufs[bv["buffer"]], dtype=dt, offset=off, count=a["count"] * ncomp).reshape(a["count"], ncomp)
        return arr

    positions, normals, indices, material_ids, materials = [], [], [], [], []
    root = np.diag([scale, scale, scale, 1.0])

    def walk(ni, xf):
        n = g["nodes"][ni]
        xf = xf @ node_matrix(n)
        if "mesh" in n:
            flip = np.linalg.det(xf) < 0
            for prim in g["meshes"][n["mesh"]]["primitives"]:
                mat = g["materials"][prim["material"]]
                pbr = mat.get("pbrMetallicRoughness", {})
                mat_index = len(materials)
                materials.append(dict(
                    base_color=pbr.get("baseColorFactor", [1, 1, 1, 1]),
                    roughness=pbr.get("roughnessFactor", 1.0), metalness=pbr.get("metallicFactor", 1.0),
                    emissive=mat.get("emissiveFactor", [0, 0, 0])))
                pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
                nrm = accessor(prim["attributes"]["NORMAL"]).astype(np.float32)
                idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1).copy()
                if flip:
                    t = idx.reshape(-1, 3)
                    t[:, [0, 2]] = t[:, [2, 0]]
                    idx = t.reshape(-1)
                base_index = sum(len(p) for p in positions)
                indices.append(idx + base_index)
                material_ids.append(np.full(len(pos), mat_index, np.uint32))
                # glam f32 math: (xform * v.extend(1)).truncate()
                xf32 = xf.astype(np.float32)
                p = (xf32[:3, :3] @ pos.T).T + xf32[:3, 3]
                nn = (xf32[:3, :3] @ nrm.T).T
                nn = nn / np.linalg.norm(nn, axis=1, keepdims=True)
                positions.append(p.astype(np.float32))
                normals.append(nn.astype(np.float32))
        for c in n.get("children", []):
            walk(c, xf)

    for ni in g["scenes"][g.get("scene", 0)]["nodes"]:
        walk(ni, root)
    return dict(
        positions=np.concatenate(positions), normals=np.concatenate(normals),
        indices=np.concatenate(indices), material_ids=np.concatenate(material_ids),
        mat_base_color=np.array([m["base_color"] for m in materials], np.float32),
        mat_roughness=np.array([m["roughness"] for m in materials], np.float32),
        mat_metalness=np.array([m["metalness"] for m in materials], np.float32),
        mat_emissive=np.array([m["emissive"] for m in materials], np.float32))


def main():
    from PIL import Image
    os.makedirs(OUT, exist_ok=True)
    im = np.array(Image.open(os.path.join(REF, "assets/images/bluenoise/256_256/LDR_RGBA_0.png")))
    assert im.shape == (256, 256, 4) and im.dtype == np.uint8
    im.tofile(os.path.join(OUT, "bluenoise_256_rgba8.bin"))
    m = load_gltf(os.path.join(REF, "assets/meshes/cornell_box/scene.gltf"), 2.0)
    np.savez_compressed(os.path.join(OUT, "cornell_box.npz"), **m)
    print("cornell:", m["positions"].shape, m["indices"].shape, "materials", len(m["mat_roughness"]))
    print("bounds", m["positions"].min(0), m["positions"].max(0))
    m = load_gltf(os.path.join(REF, "assets/meshes/pica_pica_-_mini_diorama_01/scene.gltf"), 0.1)
    # indexed vertices repeat per node instance after baking; store what the importer produces, losslessly
    np.savez_compressed(os.path.join(OUT, "pica_diorama.npz"), **m)
    print("pica:", m["positions"].shape, m["indices"].shape, "materials", len(m["mat_roughness"]))
    print("bounds", m["positions"].min(0), m["positions"].max(0))


if __name__ == "__main__":
    main()
