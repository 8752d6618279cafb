set -u
timeout 1200 python -m pytest tests -x -q -m gpu -k "big_level or golden or other_seeds or raster_parity or stress or moving or decor" 2>&1 | tail -3
for ARGS in "--big" "" "--width 320 --height 200 --poses 8192"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/cur.so _variants/new.so _variants/socc5.so -- $ARGS --other off 2>&1
done
