import os, sys
sys.path.insert(0,'/root/repo')
os.environ["LVBGPU_PROBE_VERBOSE"]="1"
import numpy as np
from lvb_amd import api
n,m=500,50000
nwords=api.words_per_row(m)
rng=np.random.default_rng(1)
enc=rng.integers(0,2**63,size=(n,nwords),dtype=np.uint64)|np.uint64(0x1111111111111111)
ctx=api.FitchContext(enc)
for rep in range(2):
    for B,rows in ((4096,24),(2048,48),(2048,39),(2048,36),(8192,12)):
        ctx.probe_l2(B,rows,20)
ctx.close()
