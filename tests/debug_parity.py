"""Debug helper (GPU box): where do HIP and the oracle disagree?"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import rust_doom_amd as rd
from oracle import raster, wad_oracle
from util import META_PATH, ensure_wad
from test_gpu_raster_parity import sweep_poses

level, W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kinds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lv = wad_oracle.build_level(ensure_wad(), META_PATH, level)
poses = sweep_poses(lv, n, W, H)
lights = lv.lights.fill_buffer_at(0.0)
dev = rd.DeviceLevel(lv); batch = rd.Batch(dev, W, H, n); batch.enable_primitive_ids()
t = batch.render(poses, lights, kinds=kinds, timed=True); print(t)
fb = batch.read_framebuffer(); prim = batch.read_primitive_ids()
ro = raster.RasterOracle(lv)
for i in range(n):
    ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, W, H, kinds=kinds, want_prim=True)
    bad = oprim != prim[i]
    print('pose', i, 'prim mismatches', bad.sum(), 'colour mismatches', (ofb != fb[i]).sum())
    if bad.sum():
        ys, xs = np.nonzero(bad)
        print('  bbox x', xs.min(), xs.max(), 'y', ys.min(), ys.max())
        pairs = collections.Counter(zip(oprim[bad].tolist(), prim[i][bad].tolist()))
        print('  (oracle, hip) pairs:', pairs.most_common(12))
        rows = collections.Counter(ys.tolist()); print('  rows:', sorted(rows.items())[:10], '...')
        cols = collections.Counter((xs % 32).tolist()); print('  cols mod 32:', sorted(cols.items()))
