set -u
for ARGS in "--big" "" "--width 320 --height 200 --poses 8192"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/cur.so _variants/bocc4.so -- $ARGS --other off 2>&1
done
