"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE (GPU box, under rocprofv3 --pmc):
land_mask reads T*C*4 bytes once (dword per lane, coalesced); gather_cells reads the same array in
32-byte pieces (8 lanes x 4 B per row, the ring kernel's read shape) and writes it once."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xmhw_amd._lib import hip
from xmhw_amd.device import DeviceBuffer
h = hip()
T, C = 14610, 262144
ts = DeviceBuffer(4 * T * C); keep = DeviceBuffer(C)
h.synth_sst(ts.ptr, 4, T, C, C, 0, 1, 0.0, 0)
h.land_mask(ts.ptr, 4, T, C, C, 0, keep.ptr)
h.stream_sync(0)
print("bytes", 4 * T * C)
