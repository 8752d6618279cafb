#!/bin/bash
# Census of the rasteriser's sixteen-pixel bodies (raster_stats hook): bodies per pass against the most one lane needed
for ARGS in "--poses 256" "--poses 1024 --width 320 --height 200" "--big --poses 256" "--poses 64 --width 3840 --height 2160"; do
  echo "== $ARGS"; python tools/raster_stats.py $ARGS 2>&1 | grep -E "per quadrant pass|sixteen-pixel"
done
