python tools/raster_stats.py 2>&1 | grep "rdoom stats" | tail -1 | sed 's/.*one-entry/one-entry/'
python tools/raster_stats.py --width 320 --height 200 --poses 2048 2>&1 | grep "rdoom stats" | tail -1 | sed 's/.*one-entry/one-entry/'
python tools/raster_stats.py --big 2>&1 | grep "rdoom stats" | tail -1 | sed 's/.*one-entry/one-entry/'
python tools/raster_stats.py --width 3840 --height 2160 --poses 64 2>&1 | grep "rdoom stats" | tail -1 | sed 's/.*one-entry/one-entry/'
