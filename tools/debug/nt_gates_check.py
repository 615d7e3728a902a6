"""A/B of the ConvLSTM cell's gate stores: plain (product) against nt (hipvsr/lib_nt.so, built with -DRNH_GATES_NT): every value
of gates / c / h at config-2 size against float64, repeated, in both Winograd geometries (the test of tests/test_parity_r03.py on the
experimental library), then launch times of both builds.   python tools/debug/nt_gates_check.py [lib_nt.so]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, 'tests')]
from hipvsr import lib as L
name = sys.argv[1] if len(sys.argv) > 1 else 'lib_nt.so'
L.LIB_PATH = os.path.join(PKG, 'hipvsr', name)
import test_parity_r03 as t
for rep in range(6):
    for cols in ('64', '128'):
        t.test_lstm_cell_gates_config2_size_vs_float64(cols)
print(name, ': 6 x 5 launches per geometry, every gate / c / h value within 1e-4 of float64: OK')
