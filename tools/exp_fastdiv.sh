cp rust-doom_amd/librdoom_hip.so /tmp/_s.so
for V in shipped fastdiv; do
  [ $V = shipped ] && cp /tmp/_s.so rust-doom_amd/librdoom_hip.so || cp _variants/$V.so rust-doom_amd/librdoom_hip.so
  echo "== $V"; bash tools/stats_lines.sh fd_$V 2>/dev/null | grep -E "^==|cull_kernel|setup_kernel"
done
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
