import sys; sys.path.insert(0,'.')
import numpy as np
from tests.cases import make_case, rel_err
from tests.test_gpu_parity import run_das, run_oracle
from oracle import das_ref as R
case = make_case(seq="FSA", interp="linear", seed=34, N=16, I1=70, I2=16)
t0 = (case["t0"] + np.arange(16) / case["fs"]).astype(np.float32).astype(np.float64)
o,p=run_das(case, kernel=2, t0=t0); r=run_oracle(case, t0=t0); g,pg=run_das(case, kernel=1, t0=t0)
print(p.kernel, p.fallback_tiles(), pg.kernel)
e=np.abs(o-r)[...,0,0]; i=np.unravel_index(np.argmax(e), e.shape); print('max err at', i, e[i], o[i], g[i], r[i], np.abs(r).max())
print('tiled vs generic', rel_err(o,g))
from tests.cases import cinv_f32
c=R.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], t0, case["fs"], cinv_f32(case["c"]), VS=True, DV=True, interp="linear", prec="double")
print('C oracle vs numpy oracle', rel_err(c.reshape(r.shape), r), 'tiled vs C', rel_err(o, c.reshape(r.shape)))
