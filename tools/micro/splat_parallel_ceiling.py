"""Times tools/micro/splat_parallel_ceiling.hip (the instruction stream a splat-parallel compositing backward cannot do without) against
the product's composite_bwd2 on the SAME frame: S-1080p-1M rendered through the product, its projected splats and tile lists taken
from ops.LAST_RASTER.  Prints both launch times (HIP events, medians of 10)."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gspl_amd  # noqa: E402,F401
from gspl_amd import _lib as L, ops, synthetic  # noqa: E402
from hip_helpers import hip_composite_fwd  # noqa: E402


def timed(fn, n=10):
    ms = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return statistics.median(ms)


def main():
    dev = torch.device("cuda:0")
    wl = synthetic.WORKLOADS["S-1080p-1M"]
    W, H = wl["width"], wl["height"]
    means, scales, quats, opac, shs = synthetic.scene(wl["n"], seed=42)
    cam = synthetic.camera(W, H, wl["fx"])
    ops.KEEP_LAST_RASTER = True
    settings = ops.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                                 scale_modifier=1.0, viewmatrix=cam["world_to_camera"].to(dev), projmatrix=cam["full_projection"].to(dev),
                                                 sh_degree=3, campos=cam["camera_center"].to(dev))
    m = means.to(dev)
    with torch.no_grad():
        ops.GaussianRasterizer(settings)(means3D=m, means2D=torch.zeros_like(m), opacities=opac.to(dev), shs=shs.to(dev), scales=scales.to(dev), rotations=quats.to(dev))
    last = ops.LAST_RASTER
    xy, con, col, op = [last[k].contiguous() for k in ("means2d", "conics", "colors", "opacities")]
    flat, offs = last["flatten_ids"].contiguous(), last["offsets"].contiguous()
    N, nI = xy.shape[0], flat.shape[0]
    bg = torch.zeros(3, device=dev)
    _, _, final_T, last_ids = hip_composite_fwd(L.GSPL_MODE_INRIA, xy, con, col, op.reshape(-1), bg, W, H, offs, flat, layout=L.GSPL_LAYOUT_CHW)
    v_out = torch.randn(3, H, W, device=dev)
    packed = torch.zeros(N, 9, device=dev)
    tw, th = (W + 15) // 16, (H + 15) // 16
    lib = L.lib()

    def product():
        packed.zero_()
        L.check(lib.gspl_composite_bwd_packed(N, nI, 3, L.GSPL_MODE_INRIA, L.GSPL_LAYOUT_CHW, L.ptr(xy), L.ptr(con), L.ptr(col), L.ptr(op), L.ptr(bg), W, H, 16, tw, th,
                                              L.ptr(offs), L.ptr(flat), L.ptr(final_T), L.ptr(last_ids), L.ptr(v_out), None, L.ptr(packed), 9, 0, None, L.stream()), "bwd")
    exp = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsplatpar.so"))
    exp.splat_parallel_ceiling.restype = ctypes.c_int
    exp.splat_parallel_ceiling.argtypes = [ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 7

    def ceiling():
        packed.zero_()
        rc = exp.splat_parallel_ceiling(N, nI, L.ptr(xy), L.ptr(con), L.ptr(col), L.ptr(op), W, H, L.ptr(offs), L.ptr(flat), L.ptr(final_T), L.ptr(last_ids),
                                        L.ptr(v_out), L.ptr(packed), L.stream())
        assert rc == 0, rc
    zero = timed(lambda: packed.zero_())
    product(); ceiling(); torch.cuda.synchronize()
    p, c = timed(product), timed(ceiling)
    processed = int((last_ids.max() > 0)) and int(torch.clamp(last_ids.view(-1), min=0).max())
    print(f"S-1080p-1M: {nI} list entries; clear of the packed rows {zero:.3f} ms (included in both)")
    print(f"composite_bwd2 (product, gspl_composite_bwd_packed): {p:.3f} ms")
    print(f"splat-parallel ceiling (lane = list entry, scans across lanes, T checkpoints assumed): {c:.3f} ms  = {c / p:.2f} x the product kernel")


if __name__ == "__main__":
    main()
