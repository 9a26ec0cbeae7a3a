#!/bin/bash
# bench.py's one-process-per-GPU forms on ONE rank, started the way the driver starts the SCALE run: a plain `python bench.py --gpus N`
# (no launcher) - bench.py launches the ranks itself.  YCGE_BENCH_FORCE_TILED=1 takes the tiled forms with a world of one (a one-GPU box).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
show='
import sys, json
l = sys.stdin.read().strip().splitlines()
try:
    d = json.loads(l[-1]); print(d["value"], "Mrays/s", d["ms_per_step"], "ms/frame;", d["config"]["form"], "rccl_world", d.get("rccl_world"), "latency_frames", d.get("latency_frames"),
        "| forms_tried", [(t["form"], t["rc"], t["seconds"]) for t in d.get("forms_tried", [])], "| batched", (d.get("batched") or {}).get("ms_per_step"),
        "| roofline", (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("mean_launch_ms"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("FAILED", e, l[-1][-600:] if l else "")'
for spec in "auto" "rccl" "resident --batch 3 --steps 51"; do
  echo "== python bench.py --gpus 1 --form $spec   (self-launch, world 1)"
  YCGE_BENCH_FORCE_TILED=1 timeout 900 python bench.py --gpus 1 --steps 52 --warmup 5 --form $spec --no-cpu-baseline --no-post 2>&1 | grep -v amdgpu.ids | python -c "$show"
done
echo "== the same under an outer launcher (what the task statement's command line does): torch.distributed.run -> bench.py --form auto"
YCGE_BENCH_FORCE_TILED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 52 --warmup 5 --no-cpu-baseline --no-post 2>&1 | grep -v amdgpu.ids | python -c "$show"
