#!/bin/bash
# Round 4, GPU call 1: reproduce GPUTEST_r03's red leg with round 3's point-to-point layer, then the product layer 20x;
# RCCL smoke (incl. p2p on RCCL); two nccl ranks on one device (expected refusal, text recorded); the N-rank tests.
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
{
echo "== round-3 layer (tools/dist_training_check.py --p2p-direct), every slice on the p2p path"
timeout 900 python tools/p2p_repeat.py --repeat 10 --mode subframes --direct
echo "== product layer (sharding._p2p)"
timeout 1500 python tools/p2p_repeat.py --repeat 20 --mode subframes
timeout 600 python tools/p2p_repeat.py --repeat 6 --mode views
} > $OUT/p2p_repeat.log 2>&1
timeout 600 python tools/rccl_smoke.py > $OUT/rccl_smoke.log 2>&1; echo "rccl_smoke rc $?" >> $OUT/rccl_smoke.log
DGS_DIST_BACKEND=nccl DGS_DIST_ONE_DEVICE=1 DGS_DIST_TIMEOUT_S=60 timeout 200 python tools/dist_training_check.py --ranks 2 --mode views --iters 8 > $OUT/nccl_two_ranks_one_device.log 2>&1; echo "rc $?" >> $OUT/nccl_two_ranks_one_device.log
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q -k "two_ranks or rccl or captured_step or graph_replay or makes_up" > $OUT/gpu_train_dist.log 2>&1; echo "rc $?" >> $OUT/gpu_train_dist.log
tail -5 $OUT/p2p_repeat.log $OUT/rccl_smoke.log $OUT/nccl_two_ranks_one_device.log $OUT/gpu_train_dist.log
