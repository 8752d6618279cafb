#!/bin/bash
# Section timers of raster_wave_kernel (shares of a wave's time per section, stderr of the library).  Build the variant first:
#   bash tools/variant.sh rtimers raster "-DRDOOM_RASTER_TIMERS -fno-slp-vectorize"      (GPU box, repo root, through gpurun)
cp rust-doom_amd/librdoom_hip.so /tmp/_s.so; cp _variants/rtimers.so rust-doom_amd/librdoom_hip.so
for ARGS in "--poses 256" "--poses 2048 --width 320 --height 200" "--big --poses 256" "--poses 64 --width 3840 --height 2160"; do
  echo "== $ARGS"; python bench.py $ARGS --streams 1 --steps 1 --warmup 0 --cpu-sample 0 --other off 2>&1 | grep "raster timers" | tail -1
done
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
