"""cProfile of the host side of the MOR demo's training loop (examples/operator_1dtmor.py): where the per-epoch time
outside the kernels goes.   python tools/mor_profile.py [epochs]"""
import cProfile, os, pstats, sys
sys.path.insert(0, '.')
sys.argv = [sys.argv[0], 'gpurun_out/op1dtmor_prof', sys.argv[1] if len(sys.argv) > 1 else '60']
sys.path.insert(0, 'examples')
import operator_1dtmor as demo
pr = cProfile.Profile()
pr.enable(); demo.main(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28); st.sort_stats('tottime').print_stats(22)
