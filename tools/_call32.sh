set -u
for ARGS in "" "--big" "--width 320 --height 200 --poses 8192"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/r64.so _variants/r0.so _variants/r16.so _variants/r32.so _variants/l32.so _variants/l128.so -- $ARGS --other off --streams 1 2>&1
done
