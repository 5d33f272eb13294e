import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from util import golden, sub
from test_gpu_parity import base_engine
eng, g, ph = base_engine('cyl')
xc, tc, fc = eng.find_boundary(g['ray_x'], g['ray_v'])
hit = fc >= 0
bad = np.nonzero(hit & (np.abs(tc - g['ray_tc']) > 1e-9 * np.abs(g['ray_tc'])))[0]
print('n bad', bad.size, 'of', hit.sum(), 'S', g['subvol_center'].shape)
for i in bad[:10]:
    print(i, g['ray_x'][i], g['ray_v'][i], 'got', tc[i], fc[i], 'want', g['ray_tc'][i], g['ray_fc'][i])
from util import rel_err
print('rel_err tc', rel_err(tc[hit], g['ray_tc'][hit]), 'max abs xc err', np.abs(xc[hit] - g['ray_xc'][hit]).max())
