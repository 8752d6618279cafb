"""ORACLE package -- test infrastructure only (see oracle/wad_oracle.py and oracle/raster_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
