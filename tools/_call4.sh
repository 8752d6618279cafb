cp rust-doom_amd/librdoom_hip.so /tmp/_ship.so; cp _variants/fstats.so rust-doom_amd/librdoom_hip.so
python bench.py --streams 1 --steps 1 --warmup 0 --poses 256 --cpu-sample 0 > /tmp/o.txt 2>&1; echo rc=$?
grep -a "frag stats" /tmp/o.txt | tail -3; tail -3 /tmp/o.txt | cut -c1-600
cp /tmp/_ship.so rust-doom_amd/librdoom_hip.so
