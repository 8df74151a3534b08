#!/bin/bash
# multi-rank code paths at HEAD on one GPU: sharded + replicated with 2 gloo ranks sharing the device; one-rank RCCL group; smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for par in sharded replicated; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --share-device --dist-backend gloo --parallelism $par --no-cpu-baseline --loop none 2>&1 | tail -1 | cut -c1-600 > $O/r06f_2ranks_shared_$par.json
python - <<PY
import json
try:
    d = json.loads(open("$O/r06f_2ranks_shared_$par.json").read())
    print("$par", d["n_gpus"], d["ms_per_step"], d["value"], d["config"].get("parallelism"))
except Exception as e:
    print("$par ERR", e, open("$O/r06f_2ranks_shared_$par.json").read()[:400])
PY
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 5 --parallelism sharded --init-dist --no-cpu-baseline --loop none 2>&1 | tail -1 | cut -c1-400
