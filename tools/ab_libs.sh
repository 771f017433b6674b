# A/B of library builds over bench configs: tools/ab_loss.sh "<lib|default> ..." "<cfg> ..."
for lib in $1; do
  for cfg in $2; do
    if [ "$lib" != "default" ]; then export DGS_LIB_PATH=$PWD/$lib; else unset DGS_LIB_PATH; fi
    python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$cfg', d['value'], d['ms_per_step'])"
  done
done
