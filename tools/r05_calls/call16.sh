#!/bin/bash
# r05 call 16: the N-rank code path with EIGHT ranks on the one GPU (gloo; functional only): both sharding modes, chunked
# reduction, captured fronts, the point-to-point reduce-scatter with seven peers, ranks without subframes (K = 5 < 8),
# bench.py --gpus 8 in both launch forms
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 DGS_DIST_TIMEOUT_S=600
{
for mode in views subframes; do
  echo "== $mode, collective, 4 chunks, captured front"; timeout 1500 python tools/dist_training_check.py --ranks 8 --mode $mode --ar-chunks 4 --graph always --random-sample --iters 24 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
  echo "== $mode, point-to-point on every slice"; DGS_DIST_ALLREDUCE=p2p DGS_DIST_P2P_MIN_NUMEL=0 timeout 1500 python tools/dist_training_check.py --ranks 8 --mode $mode --ar-chunks 4 --iters 24 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -3
done
echo "== bench.py --gpus 8 (self-launch), cfg2"
timeout 1500 python bench.py --gpus 8 --config cfg2 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['per_rank'], d['extras']['other_mode'], {k:(v if not isinstance(v,dict) else v.get('ms')) for k,v in d['extras']['allreduce_ab'].items() if k!='note'})"
echo "== torchrun form, 8 ranks"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --config cfg2 --steps 4 --warmup 2 --no-cpu-baseline --shard subframes 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['value'], d['extras']['other_mode'].get('value'))"
} 2>&1 | tee $OUT/c16_eight_ranks.log
