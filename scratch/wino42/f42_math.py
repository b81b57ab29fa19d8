"""F(4,2) Cook-Toom matrices (points 0, 1, -1, 2, inf) checked against the direct correlation, 1-D and 2-D,
and the phase decomposition of the 4x4 / stride-2 convolution and of its transpose that conv_wino42.hip uses."""
import numpy as np
from fractions import Fraction as Fr
AT = np.array([[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,0],[0,1,-1,8,1]], dtype=np.float64)
G = np.array([[.5,0],[-.5,-.5],[-1/6,1/6],[1/6,1/3],[0,1]], dtype=np.float64)
BT = np.array([[2,-1,-2,1,0],[0,-2,-1,1,0],[0,2,-3,1,0],[0,-1,0,1,0],[0,2,-1,-2,1]], dtype=np.float64)
rng = np.random.default_rng(0)
d = rng.standard_normal(5); g = rng.standard_normal(2)
y = AT @ ((G @ g) * (BT @ d))
ref = np.array([g[0]*d[u] + g[1]*d[u+1] for u in range(4)])
print("1D err", np.abs(y-ref).max())
d2 = rng.standard_normal((5,5)); g2 = rng.standard_normal((2,2))
Y = AT @ ((G @ g2 @ G.T) * (BT @ d2 @ BT.T)) @ AT.T
R = np.array([[sum(g2[a,b]*d2[u+a,v+b] for a in range(2) for b in range(2)) for v in range(4)] for u in range(4)])
print("2D err", np.abs(Y-R).max())
