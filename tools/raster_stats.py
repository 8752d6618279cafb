import sys, numpy as np, importlib
sys.path.insert(0,'/root/repo')
import rust_doom_amd as rd
sharding = importlib.import_module('rust-doom_amd.sharding'); syn = importlib.import_module('rust-doom_amd.synthetic')
rd.set_device(0)
built = rd.Wad(syn.ensure_wad(), syn.META_PATH).build_level(0, gpu_tessellation=True)
level = rd.DeviceLevel(built); n=256
batch = rd.Batch(level, 1920, 1080, n)
poses = sharding.pose_sweep(rd, built, n, 1920, 1080); lights = built.lights_at(0.0)
batch.render(poses, lights, timed=True)
rd.debug_set('raster_stats', 1)
print(batch.render(poses, lights, timed=True))
