#!/bin/bash
# round 6 call 16: COUNT fused into preprocess (default build) against its own launch (variants/libdgs_nofuse.so, -DDGS_FUSE_CULL=0):
# bit identity of the binning state, the tile_cull / binning tests, step time interleaved
mkdir -p gpurun_out/r06
L=gpurun_out/r06/fuse_cull.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cull or binning or forward" 2>&1 | tail -3 >> $L
for rep in 1 2 3; do
  echo -n "fused     " >> $L; timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
  echo -n "own launch" >> $L; DGS_LIB_PATH=variants/libdgs_nofuse.so timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
done
echo -n "fused sh3 " >> $L; timeout 600 python bench.py --sh-degree 3 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
echo -n "own   sh3 " >> $L; DGS_LIB_PATH=variants/libdgs_nofuse.so timeout 600 python bench.py --sh-degree 3 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
for cfg in cfg2 metric; do
  python tools/grad_hash.py $cfg > /tmp/h_fused_$cfg.txt 2>&1
  DGS_LIB_PATH=variants/libdgs_nofuse.so python tools/grad_hash.py $cfg > /tmp/h_own_$cfg.txt 2>&1
  if cmp -s /tmp/h_fused_$cfg.txt /tmp/h_own_$cfg.txt; then echo "$cfg: every output bit-identical ($(wc -l < /tmp/h_fused_$cfg.txt) hashes)" >> $L; else echo "$cfg: OUTPUTS DIFFER" >> $L; diff /tmp/h_fused_$cfg.txt /tmp/h_own_$cfg.txt >> $L; fi
done
cat $L
