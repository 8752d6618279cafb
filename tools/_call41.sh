set -u
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for ARGS in "--width 320 --height 200 --poses 8192" "--big" "" "--width 1280 --height 720 --poses 2048"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/cur.so _variants/new.so -- $ARGS --other off 2>&1
done
