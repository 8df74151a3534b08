"""`python -m gspl_amd.launch <script> [args...]` — run one of the reference's entry points (main.py, viewer.py, utils/*.py) on a
machine that has no `diff_gaussian_rasterization` / `gsplat` / `simple_knn` / `fused_ssim` (they are CUDA packages).

The reference imports `diff_gaussian_rasterization` while `internal.renderers` is imported (internal/renderers/vanilla_renderer.py:14),
i.e. before its CLI has parsed `--model.renderer`: the stand-ins of `gspl_amd.compat` therefore have to be registered BEFORE the
entry point is imported.  This launcher does exactly that and then runs the script unchanged, as `python <script> [args...]` would:

    python -m gspl_amd.launch main.py fit --data.path data/lego --model.renderer gspl_amd.renderers.HipVanillaRenderer
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        sys.exit("usage: python -m gspl_amd.launch <script.py> [args...]")
    from . import compat
    compat.install()
    script = argv[0]
    sys.argv = argv
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))      # what `python <script>` puts first
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
