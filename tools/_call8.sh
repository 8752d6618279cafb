set -u
export TMPDIR=/tmp; R=$PWD
cp rust-doom_amd/librdoom_hip.so /tmp/_ship.so
for v in shipped qocc7 qocc6 qocc5; do
  [ $v = shipped ] && cp /tmp/_ship.so rust-doom_amd/librdoom_hip.so || cp _variants/$v.so rust-doom_amd/librdoom_hip.so
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04f/$v -o r --output-format csv -- python $R/bench.py --streams 1 --steps 4 --warmup 1 --cpu-sample 0 > /dev/null 2>&1)
  echo "== $v"; grep -h "fragment_quadrant\|fragment_kernel" $(find gpurun_out/r04f/$v -name '*kernel_stats.csv') | sed 's/(.*)",/",/' | cut -c1-120
done
cp /tmp/_ship.so rust-doom_amd/librdoom_hip.so
