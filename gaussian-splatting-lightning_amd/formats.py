"""Checkpoint / PLY formats of the Gaussian model, and the merge of per-rank checkpoints of the sharded trainer
(reference `internal/utils/gaussian_utils.py:52-255` `GaussianPlyUtils`,
`utils/merge_distributed_ckpts.py`).  No third-party dependency (the reference goes through `plyfile`): the PLY codec below
reads and writes the binary_little_endian / ascii vertex element the 3DGS ecosystem uses.

Layouts:
  model / checkpoint  means [N,3], shs_dc [N,1,3], shs_rest [N,K-1,3] (coefficient-major, channel-minor), opacities [N,1]
                      (logits), scales [N,3] (logs), rotations [N,4] (wxyz, unnormalised); state-dict keys
                      `gaussian_model.gaussians.<name>` (older checkpoints: `gaussian_model._xyz`, `_features_dc`, ...).
  PLY                 x y z | f_dc_0..2 | f_rest_0..3(K-1)-1 | opacity | scale_0..2 | rot_0..3, float32; `f_rest_*` is
                      CHANNEL-major ([N,3,K-1] flattened): transposed on load and save (gaussian_utils.py:63-67,168-201).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

SHS_REST_DIM_TO_DEGREE = {0: 0, 3: 1, 8: 2, 15: 3, 24: 4}
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
_NP_TO_PLY = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint"}


# ---------------------------------------------------------------------------------------------------------------------
# PLY codec (vertex element only)
# ---------------------------------------------------------------------------------------------------------------------
def read_ply_vertices(path: str) -> np.ndarray:
    """The `vertex` element of a PLY file as a numpy structured array (binary little/big endian or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, elements = None, 0, [], False, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                elements.append(tok[1])
                if in_vertex:
                    count = int(tok[2])
                    if len(elements) != 1:
                        raise ValueError(f"{path}: the vertex element must come first")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or not props:
            raise ValueError(f"{path}: no format / vertex properties")
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for i, (n, _) in enumerate(props):
                out[n] = data[:, i]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        raw = f.read(count * dt.itemsize)
        if len(raw) != count * dt.itemsize:
            raise ValueError(f"{path}: truncated vertex data")
        return np.frombuffer(raw, dtype=dt, count=count)


def write_ply_vertices(path: str, vertices: np.ndarray) -> None:
    """Write a structured array as the `vertex` element of a binary_little_endian PLY (the header `plyfile` writes)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    names = vertices.dtype.names
    le = np.dtype([(n, "<" + vertices.dtype[n].str[1:]) for n in names])
    body = np.ascontiguousarray(vertices.astype(le, copy=False))
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {vertices.shape[0]}"]
    header += [f"property {_NP_TO_PLY[vertices.dtype[n].str[1:]]} {n}" for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(body.tobytes())


# ---------------------------------------------------------------------------------------------------------------------
# Gaussian model <-> PLY / state dict
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class GaussianProperties:
    """Model-layout tensors (float32): means [N,3], shs_dc [N,1,3], shs_rest [N,K-1,3], opacities [N,1], scales [N,3],
    rotations [N,4] — raw (pre-activation) values, as stored in checkpoints and PLY files."""
    means: torch.Tensor
    shs_dc: torch.Tensor
    shs_rest: torch.Tensor
    opacities: torch.Tensor
    scales: torch.Tensor
    rotations: torch.Tensor

    NAMES = ("means", "shs_dc", "shs_rest", "opacities", "scales", "rotations")

    @property
    def sh_degree(self) -> int:
        return SHS_REST_DIM_TO_DEGREE[int(self.shs_rest.shape[1])]

    def as_dict(self) -> Dict[str, torch.Tensor]:
        return {k: getattr(self, k) for k in self.NAMES}

    # ---- state dicts (gaussian_utils.py:113-159) ----
    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor]) -> "GaussianProperties":
        if "gaussian_model.gaussians.means" in state_dict:
            p = "gaussian_model.gaussians."
            return cls(**{k: state_dict[p + k].detach().float().cpu() for k in cls.NAMES})
        p = "gaussian_model._"          # checkpoints written before the property-dict model
        old = {"means": "xyz", "shs_dc": "features_dc", "shs_rest": "features_rest", "opacities": "opacity", "scales": "scaling", "rotations": "rotation"}
        return cls(**{k: state_dict[p + v].detach().float().cpu() for k, v in old.items()})

    @classmethod
    def from_model(cls, model) -> "GaussianProperties":
        props = model.properties
        return cls(**{k: props[k].detach().float().cpu() for k in cls.NAMES})

    def to_state_dict(self, prefix: str = "gaussian_model.gaussians.") -> Dict[str, torch.Tensor]:
        return {prefix + k: v for k, v in self.as_dict().items()}

    # ---- PLY (gaussian_utils.py:52-86,163-255) ----
    @classmethod
    def load_ply(cls, path: str, sh_degree: int = -1) -> "GaussianProperties":
        v = read_ply_vertices(path)
        names = v.dtype.names
        n = v.shape[0]

        def cols(prefix, required=True):
            found = sorted([k for k in names if k.startswith(prefix)], key=lambda k: int(k.split("_")[-1]))
            if not found:
                if required:
                    raise RuntimeError(f"'{prefix}' not found in ply")
                return np.empty((n, 0), np.float32)
            return np.stack([np.asarray(v[k], np.float32) for k in found], axis=1)

        means = np.stack([np.asarray(v[k], np.float32) for k in ("x", "y", "z")], axis=1)
        dc = cols("f_dc_").reshape(n, 3, 1)                       # channel-major in the file
        rest = cols("f_rest_", required=False).reshape(n, 3, -1)
        if sh_degree >= 0 and rest.shape[-1] != (sh_degree + 1) ** 2 - 1:
            raise ValueError(f"ply holds {rest.shape[-1]} rest coefficients per channel, sh_degree {sh_degree} needs {(sh_degree + 1) ** 2 - 1}")
        if rest.shape[-1] not in SHS_REST_DIM_TO_DEGREE:
            raise ValueError(f"invalid number of SH rest coefficients: {rest.shape[-1]}")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return cls(means=t(means), shs_dc=t(dc.transpose(0, 2, 1)), shs_rest=t(rest.transpose(0, 2, 1)),
                   opacities=t(np.asarray(v["opacity"], np.float32)[:, None]), scales=t(cols("scale_")), rotations=t(cols("rot_")))

    def save_ply(self, path: str, with_colors: bool = False) -> None:
        n = self.means.shape[0]
        npy = lambda t: t.detach().float().cpu().numpy()
        f_dc = npy(self.shs_dc).transpose(0, 2, 1).reshape(n, -1)                     # -> [N,3,1] channel-major
        f_rest = npy(self.shs_rest).transpose(0, 2, 1).reshape(n, -1) if self.shs_rest.shape[1] > 0 else np.zeros((n, 0), np.float32)
        fields = [("x", "f4"), ("y", "f4"), ("z", "f4")]
        values = [npy(self.means)[:, 0], npy(self.means)[:, 1], npy(self.means)[:, 2]]

        def add(prefix, a):
            for i in range(a.shape[1]):
                fields.append((f"{prefix}_{i}", "f4"))
                values.append(a[:, i])
        add("f_dc", f_dc)
        add("f_rest", f_rest)
        fields.append(("opacity", "f4"))
        values.append(npy(self.opacities).reshape(n))
        add("scale", npy(self.scales))
        add("rot", npy(self.rotations))
        if with_colors:      # degree-0 colour as uchar red/green/blue (gaussian_utils.py:236-246)
            rgb = np.clip(0.28209479177387814 * npy(self.shs_dc).reshape(n, 3) + 0.5, 0.0, 1.0)
            rgb = (rgb * 255).astype(np.uint8)
            for i, c in enumerate(("red", "green", "blue")):
                fields.append((c, "u1"))
                values.append(rgb[:, i])
        out = np.empty(n, dtype=fields)
        for (name, _), val in zip(fields, values):
            out[name] = val
        write_ply_vertices(path, out)


# ---------------------------------------------------------------------------------------------------------------------
# per-rank checkpoints of the Gaussian-sharded trainer -> one checkpoint (utils/merge_distributed_ckpts.py)
# ---------------------------------------------------------------------------------------------------------------------
def merge_rank_checkpoints(ckpts: Sequence[dict]) -> dict:
    """One checkpoint from the per-rank checkpoints of the Gaussian-sharded trainer (utils/merge_distributed_ckpts.py), in rank
    order.  Returns a NEW top-level dict (the inputs are not modified; tensors that are not concatenated are shared):

      * `gaussian_model.gaussians.*` and `density_controller.*` of the state dict are concatenated along the Gaussian axis;
      * the Adam moments (`exp_avg`, `exp_avg_sq`) of the FIRST optimizer group named after each Gaussian property are concatenated
        (the reference stops at the first match per property: a second optimizer that happens to reuse a name is left alone);
      * the renderer hyper-parameter, when it is a distributed renderer, is replaced by the non-distributed one carrying over
        block_size / anti_aliased / filter_2d_kernel_size / tile_based_culling (`HipGSplatV1Renderer`; the reference swaps in its
        GSplatV1Renderer), and `renderer.appearance_model.module.*` (DDP-wrapped appearance model) is renamed `renderer.model.*`;
      * everything else comes from the last checkpoint.
    The pre-property-dict layout (`gaussian_model._xyz` + `gaussian_model_extra_state_dict`) is not written by any renderer of this
    package and is rejected."""
    if not ckpts:
        raise ValueError("no checkpoints")
    gp, dp = "gaussian_model.gaussians.", "density_controller."
    last = ckpts[-1]
    if not any(k.startswith(gp) for k in last["state_dict"]):
        if "gaussian_model._xyz" in last["state_dict"] or "gaussian_model_extra_state_dict" in last:
            raise ValueError("checkpoint in the pre-property-dict layout (gaussian_model._xyz / gaussian_model_extra_state_dict): "
                             "merge it with the reference's utils/merge_distributed_ckpts.py")
        raise ValueError("no `gaussian_model.gaussians.*` entries in the state dict")
    merged = dict(last)
    sd = dict(last["state_dict"])
    names = []
    for k in list(sd):
        if k.startswith(gp) or k.startswith(dp):
            sd[k] = torch.cat([c["state_dict"][k] for c in ckpts], dim=0)
            if k.startswith(gp):
                names.append(k[len(gp):])
    ddp_prefix = "renderer.appearance_model.module."
    for k in [k for k in sd if k.startswith(ddp_prefix)]:
        sd["renderer.model." + k[len(ddp_prefix):]] = sd.pop(k)
    merged["state_dict"] = sd

    pending = list(names)
    optimizers = []
    for oi, opt in enumerate(last.get("optimizer_states", [])):
        opt = dict(opt)
        state = dict(opt.get("state", {}))
        for gi, group in enumerate(opt.get("param_groups", [])):
            name = group.get("name")
            if name not in pending or gi not in state:
                continue
            pending.remove(name)
            entry = dict(state[gi])
            for m in ("exp_avg", "exp_avg_sq"):
                entry[m] = torch.cat([c["optimizer_states"][oi]["state"][gi][m] for c in ckpts], dim=0)
            state[gi] = entry
        opt["state"] = state
        optimizers.append(opt)
    if "optimizer_states" in last:
        merged["optimizer_states"] = optimizers

    hp = last.get("hyper_parameters")
    renderer = hp.get("renderer") if isinstance(hp, dict) else None
    kind = type(renderer).__name__ if renderer is not None else ""
    if kind in ("GSplatDistributedRenderer", "HipGSplatDistributedRenderer"):
        # the plain Gaussian-sharded renderers -> the single-process plugin (utils/merge_distributed_ckpts.py:130-141)
        from .renderers import HipGSplatV1Renderer
        hp = dict(hp)
        hp["renderer"] = HipGSplatV1Renderer(
            block_size=getattr(renderer, "block_size", 16), anti_aliased=getattr(renderer, "anti_aliased", True),
            filter_2d_kernel_size=getattr(renderer, "filter_2d_kernel_size", 0.3),
            separate_sh=getattr(renderer, "separate_sh", False),
            tile_based_culling=getattr(renderer, "tile_based_culling", False))
        merged["hyper_parameters"] = hp
    elif "Distributed" in kind:
        # GSplatDistributedAppearanceEmbedding(Mip)Renderer (utils/merge_distributed_ckpts.py:143-170) carry an appearance model
        # whose weights were renamed `renderer.model.*` above; their single-process counterparts are not part of this package
        # (appearance models are out of scope), and swapping in a plain renderer would drop the model and leave
        # orphan keys for a strict load
        raise NotImplementedError(f"merge_rank_checkpoints: renderer {kind} has an appearance model; merge this checkpoint with the "
                                  "reference's utils/merge_distributed_ckpts.py (it maps it to GSplatAppearanceEmbedding(Mip)Renderer)")
    return merged


def find_rank_checkpoints(checkpoint_dir: str) -> List[str]:
    """The `...step=<S>-rank=<R>.ckpt` files of the highest step, ordered by rank."""
    best, files = -1, []
    for name in os.listdir(checkpoint_dir):
        if not name.endswith(".ckpt") or "step=" not in name or "-rank=" not in name:
            continue
        try:
            step = int(name[name.index("step=") + 5:name.rindex("-rank=")])
            rank = int(name[name.rindex("-rank=") + 6:-5])
        except ValueError:
            continue
        if step > best:
            best, files = step, []
        if step == best:
            files.append((rank, name))
    return [os.path.join(checkpoint_dir, n) for _, n in sorted(files)]


def main(argv: Optional[Sequence[str]] = None) -> None:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m gspl_amd.formats")
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("merge", help="merge the per-rank checkpoints of <model output dir>/checkpoints")
    m.add_argument("path")
    c = sub.add_parser("ckpt2ply", help="write the Gaussians of a checkpoint as a PLY file")
    c.add_argument("ckpt")
    c.add_argument("ply")
    c.add_argument("--colors", action="store_true")
    p = sub.add_parser("ply2ckpt", help="state dict (.pt) with the Gaussians of a PLY file")
    p.add_argument("ply")
    p.add_argument("out")
    a = ap.parse_args(argv)
    if a.cmd == "merge":
        d = os.path.join(a.path, "checkpoints")
        files = find_rank_checkpoints(d)
        assert files, f"no per-rank checkpoints in {d}"
        merged = merge_rank_checkpoints([torch.load(f, map_location="cpu", weights_only=False) for f in files])
        out = files[0][:files[0].rindex("-rank=")] + ".ckpt"
        torch.save(merged, out)
        print(f"{len(files)} checkpoints -> {out} ({merged['state_dict']['gaussian_model.gaussians.means'].shape[0]} Gaussians)")
    elif a.cmd == "ckpt2ply":
        ckpt = torch.load(a.ckpt, map_location="cpu", weights_only=False)
        GaussianProperties.from_state_dict(ckpt["state_dict"]).save_ply(a.ply, with_colors=a.colors)
    else:
        torch.save(GaussianProperties.load_ply(a.ply).to_state_dict(), a.out)


if __name__ == "__main__":
    main()
