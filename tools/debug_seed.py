import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import lasgun_amd as la
from oracle_lib import oracle
S = la.scenes; G = la.api; o = oracle()
seed = int(sys.argv[1]); w, h = 56, 40
oacc = o.Accel(S.random_scene(o, seed)); o.set_trig_mode(1); orad = o.capture_radiance(oacc, w, h, nthreads=8); o.set_trig_mode(0)
acc = G.Accel(S.random_scene(G, seed)); G.set_streaming(acc, 0)
grad = G.capture_radiance(acc, w, h)
d = np.argwhere(grad.view(np.uint64) != orad.view(np.uint64))
print("ndiff", len(d))
for y, x, c in d[:12]:
    print(y, x, c, repr(grad[y, x, c]), repr(orad[y, x, c]), hex(grad.view(np.uint64)[y, x, c]), hex(orad.view(np.uint64)[y, x, c]))
