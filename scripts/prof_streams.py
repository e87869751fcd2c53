import sys, time, ctypes
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import oracle_lib as O
from bwtc_amd import synth, hip
import test_host_logic as T
H = hip.load()
d = synth.gen_text(32<<20, 3)
bwt, lf, freqs = O.ref_bwt_block(d, 8)
sections = O.oracle_sections(freqs)
for th in (1,15):
    t=time.time(); T._host_wavelet_payload(H, bwt, sections, 4, th, "bwtc_hip_host_wavelet_streams"); print(th, time.time()-t)
