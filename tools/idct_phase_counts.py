"""Decode one resident batch of 16 synthetic 4K images a few times (no parity check): the workload under
`rocprofv3 --pmc SQ_INSTS_VALU` for the per-phase instruction accounting of k_idct_color
(tools/idct_phase_counts.sh builds the library with -DHJD_IDCT_CUT=1/2 and points HJD_LIB at each build)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hjdtools as T

hjd = importlib.import_module("hls-jpeg-decoder_amd")
jpgs = [T.synth_jpeg(3840, 2160, 1000 + s, 85, 2, 2) for s in range(8)] * 2
dec = hjd.Decoder(0, streams=1)  # one launch of 16 images per kernel
b = dec.prepare(jpgs)
for _ in range(3):
    b.run()
    b.finish()
print("done")
