#!/bin/bash
# N > 1 code path of bench.py after this round's edits: 2 ranks sharing cuda:0, collectives over gloo (test hook of bench.py)
set +e
OUT=gpurun_out/r2s36; mkdir -p $OUT
export GPAMD_BENCH_BACKEND=gloo GPAMD_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --size 100000 > $OUT/bench_n2_metric.json 2> $OUT/bench_n2_metric.err; echo "rc=$?"; cut -c1-700 $OUT/bench_n2_metric.json; tail -3 $OUT/bench_n2_metric.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --config c4 --size 200000 > $OUT/bench_n2_c4.json 2> $OUT/bench_n2_c4.err; echo "rc=$?"; cut -c1-700 $OUT/bench_n2_c4.json; tail -3 $OUT/bench_n2_c4.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 1 --warmup 1 --size 100000 --contraction split > $OUT/bench_n2_split.json 2> $OUT/bench_n2_split.err; echo "rc=$?"; cut -c1-400 $OUT/bench_n2_split.json; tail -3 $OUT/bench_n2_split.err
