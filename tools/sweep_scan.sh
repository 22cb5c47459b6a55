for r in 4 8 12 16 24 32 48; do
  for fl in 3 0; do
    t=$(( 1 | (1<<8) | (r<<16) | (fl<<30) ))
    MIRT_PATHS_TUNING=$t timeout -k 10 120 python bench.py --config C4 --spp 128 --steps 2 --warmup 1 --cpu-seconds 0 | python -c "import sys,json; d=json.load(sys.stdin); print('refill_at $r flags $fl', round(d['value'],1))" || exit 1
  done
done
