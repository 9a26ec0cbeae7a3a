#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_variants.py -m gpu -q -x -k "tile_resident or placed_gate or waived or two_rank or tiled or texture" > gpurun_out/r4_res_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r4_res_pytest.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refill or every_form" > gpurun_out/r4_exp_pytest.log 2>&1; echo "experiments pytest rc=$?"; tail -4 gpurun_out/r4_exp_pytest.log
echo "== a rank's period, resident ring"; for k in 2 4 6; do timeout 300 python profiles/rank_flight.py 4 8 resident $k 2>&1 | tail -1; done; timeout 300 python profiles/rank_flight.py 4 8 two 2>&1 | tail -1
timeout 300 python profiles/rank_flight.py 4 4 resident 4 2>&1 | tail -1; timeout 300 python profiles/rank_flight.py 4 2 resident 4 2>&1 | tail -1; timeout 300 python profiles/rank_flight.py 3 8 resident 4 2>&1 | tail -1
echo "== bench, resident form on one rank under torchrun (smoke)"; YCGE_BENCH_FORCE_TILED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --form resident --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-900
