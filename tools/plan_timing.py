import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import symbolic as sy
from pastix_amd import dist as pd
N=int(sys.argv[1])
n,cp,r,v=sy.laplacian_3d(N)
perm,invp=sy.order_grid(N,N,N)
t=time.time(); s=sy.symbolic(n,cp,r,perm); print("symbolic %.2f s"%(time.time()-t))
t=time.time(); pd.plan_profile(s["cblk4"],s["blok4"],None,0); print("plan %.2f s"%(time.time()-t))
