import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import fixture_io, numpy as np
from pastix_amd import symbolic as sy, fact_flops
def nd(N):
    invp=[]
    def rec(x0,x1,y0,y1,z0,z1):
        dx,dy,dz=x1-x0,y1-y0,z1-z0; cnt=dx*dy*dz
        if cnt<=0: return
        if cnt<=8:
            for z in range(z0,z1):
                for y in range(y0,y1):
                    for x in range(x0,x1): invp.append(x+N*(y+N*z))
            return
        if dx>=dy and dx>=dz:
            m=x0+dx//2; rec(x0,m,y0,y1,z0,z1); rec(m+1,x1,y0,y1,z0,z1)
            for z in range(z0,z1):
                for y in range(y0,y1): invp.append(m+N*(y+N*z))
        elif dy>=dz:
            m=y0+dy//2; rec(x0,x1,y0,m,z0,z1); rec(x0,x1,m+1,y1,z0,z1)
            for z in range(z0,z1):
                for x in range(x0,x1): invp.append(x+N*(m+N*z))
        else:
            m=z0+dz//2; rec(x0,x1,y0,y1,z0,m); rec(x0,x1,y0,y1,m+1,z1)
            for y in range(y0,y1):
                for x in range(x0,x1): invp.append(x+N*(y+N*m))
    rec(0,N,0,N,0,N)
    invp=np.array(invp); perm=np.empty_like(invp); perm[invp]=np.arange(len(invp)); return perm
for name,N,bs in [('lap3d_8_llt',8,120),('rlap3d_12_llt',12,120),('rlap3d_14_llt_bs24',14,24),('rlap3d_20_llt_bs128',20,128)]:
    g=fixture_io.load_npz('tests/golden/%s.npz'%name)
    c4=g['cblk4']; w=c4[:-1,1]-c4[:-1,0]+1
    nnz_ref=int((c4[:-1,3]*w - w*(w-1)//2).sum())
    s=sy.symbolic(g['n'],g['colptr'],g['rows'],nd(N),max_blocksize=bs,amalgamation_pct=5)
    fl=fact_flops(s['cblk4'],s['blok4'],0)
    print("%-22s nnzL %+.2f%% flops %+.2f%% cblk ref %d ours %d  fund %d amalg %d"%(name,100*(s['nnzl']/nnz_ref-1),100*(fl/g['flops']-1),len(c4)-1,len(s['cblk4'])-1,s['nsuper_fund'],s['nsuper_amalg']))
