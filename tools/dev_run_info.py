import sys, ctypes, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import symbolic as sy, _lib
from pastix_amd.solver import LayoutArrays
from pastix_amd._lib import Options
N=int(sys.argv[1]); maxc=int(sys.argv[2]) if len(sys.argv)>2 else 0
n,cp,r,v=sy.laplacian_3d(N); perm,_=sy.order_grid(N,N,N)
s=sy.symbolic(n,cp,r,perm,max_blocksize=128)
la=LayoutArrays(s["cblk4"],s["blok4"])
o=Options(); o.run_max_cblks=maxc
info=(ctypes.c_int64*8)()
t=time.time()
rc=_lib.lib().pastix_amd_plan_run_info(ctypes.byref(la.c),0,1,ctypes.byref(o),info)
print("rc",rc,"L0 %d levels %d run tasks %d waits %d gd %d Ttasks %d runflops %.3e verify %d"%tuple(info), "%.2fs"%(time.time()-t))
