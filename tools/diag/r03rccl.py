import os, subprocess, sys
sys.path.insert(0, "tests")
import test_rccl_single_rank as t
open("/tmp/worker.py", "w").write(t.WORKER)
for i in range(6):
    env = dict(os.environ, GSPL_ROOT=os.getcwd(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + i), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "/tmp/worker.py"], env=env, capture_output=True, text=True)
    print("run", i, "rc", r.returncode)
    if r.returncode != 0:
        lines = [l for l in r.stderr.splitlines() if "Warning" not in l and "amdgpu.ids" not in l]
        print("\n".join(lines[-25:]))
