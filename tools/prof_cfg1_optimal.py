import cProfile, pstats, sys, tempfile, time
sys.path.insert(0, '/root/repo')
import numpy as np
from tests.test_exact_tables import _config1_problem
vn = _config1_problem()
np.random.seed(0)
pr = cProfile.Profile()
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter(); pr.enable()
    res = vn.train(tmp, weight=[10., 10., 1.], smpScheme='optimal', adjustWeight=True, epochNum=40000, verbose=False)
    pr.disable(); dt = time.perf_counter() - t0
print('40000 epochs in %.2f s, redraws %s' % (dt, res.inpIter))
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(22)
st.sort_stats('cumulative'); st.print_stats(16)
