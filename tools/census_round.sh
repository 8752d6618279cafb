set -u
{
for a in "--poses 256" "--poses 1024 --width 320 --height 200" "--big --poses 256" "--big --poses 64 --width 3840 --height 2160"; do
echo "== python tools/raster_stats.py $a"
python tools/raster_stats.py $a 2>&1 | grep "rdoom stats\|plain"
echo "-- the same with rdoom_debug_set no_split=1"
RDOOM_STATS_HOOK=no_split python tools/raster_stats.py $a 2>&1 | grep "per quadrant pass\|plain"
done
} > gpurun_out/r04_raster_census.txt 2>&1
tail -5 gpurun_out/r04_raster_census.txt | cut -c1-200
