"""CPU model of the IDCT term loop of k_idct_color (csrc/hjd_pixels.hip, phase 2): for one 3840x2160 4:2:0 q85 bench image, the
number of loop iterations a wave executes (the 16 teams of a wave walk their blocks' terms in lock step, so a wave pays for its
longest block) under different block-to-team assignments and team sizes.  Uses the oracle for the coefficients (test tooling
only).  Numbers quoted in DESIGN.md §4.2 come from here:  python tools/idct_model.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hjdtools as T
W,H=3840,2160
j=T.synth_jpeg(W,H,1000,85,2,2)
r=T.oracle_decode(j,want_coefs=True)
coefs=r[2]
coefs=np.asarray(coefs).reshape(-1,64)
print(coefs.shape)
ZZ=[0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63]
nat=np.zeros_like(coefs); 
for z,n in enumerate(ZZ): nat[:,n]=coefs[:,z]
# slot = u*8+v with u horizontal frequency: natural index n = row(v)*8+col(u) -> slot = (n%8)*8 + n//8
slot=np.zeros_like(nat)
for n in range(64): slot[:,(n%8)*8+n//8]=nat[:,n]
nz=(slot!=0)
ac=nz.copy(); ac[:,0]=False
lo=ac[:,:32].sum(1); hi=ac[:,32:].sum(1)
nb=6; mcux=W//16; mcuy=H//16
blk=np.arange(coefs.shape[0]); comp=blk%6
print("mean AC terms luma %.2f chroma %.2f all %.2f; dc nonzero %.3f"%(ac[comp<4].sum(1).mean(), ac[comp>=4].sum(1).mean(), ac.sum(1).mean(), nz[:,0].mean()))
print("mean lo %.2f hi %.2f"%(lo.mean(),hi.mean()))
# simulate
tot_iters=0; tot_terms=0; waves=0
G=16
for my in range(mcuy):
    for gx in range(mcux//G):
        b0=(my*mcux+gx*G)*6
        K=G; KB=K*6; nluma=K*4
        for s0 in range(0,KB,16):  # one wave-pass of 16 teams
            ss=np.arange(s0,min(s0+16,KB))
            bl=np.where(ss<nluma,(ss>>2)*6+(ss&3),((ss-nluma)>>1)*6+4+((ss-nluma)&1))+b0
            tot_iters+=lo[bl].max()+hi[bl].max(); tot_terms+=(lo[bl]+hi[bl]).sum(); waves+=1
nblocks=coefs.shape[0]
print("wave-pass loop iterations %d (%.2f per wave-pass), useful team-terms %d, efficiency %.3f"%(tot_iters,tot_iters/waves,tot_terms,tot_terms/(tot_iters*16)))
print("term-loop VALU wave-instr per image: %.2fM (30/iter) = %.1f per block"%(tot_iters*30/1e6, tot_iters*30/nblocks))
# sorted assignment ideal: sort all 96 blocks in group by count
tot2=0
for my in range(mcuy):
    for gx in range(mcux//G):
        b0=(my*mcux+gx*G)*6
        c=np.sort((lo+hi)[b0:b0+96])[::-1]
        l2=np.sort(lo[b0:b0+96])[::-1]
        for s0 in range(0,96,16): tot2+=c[s0]
print("sorted-by-total (single loop) iterations %d -> %.1f VALU/block"%(tot2,tot2*30/nblocks))
# variants
def run(keyfun, single):
    tot=0
    for my in range(mcuy):
        for gx in range(mcux//G):
            b0=(my*mcux+gx*G)*6
            idx=np.arange(b0,b0+96)
            order=idx[np.argsort(-keyfun(idx),kind='stable')]
            for s0 in range(0,96,16):
                bl=order[s0:s0+16]
                tot+= (lo[bl]+hi[bl]).max() if single else lo[bl].max()+hi[bl].max()
    return tot
for name,kf,single in [("sort total, two loops",lambda i:(lo+hi)[i],False),("sort total, single loop",lambda i:(lo+hi)[i],True),
                       ("sort lo, two loops",lambda i:lo[i],False),("sort quantised total/2",lambda i:((lo+hi)[i]+1)//2,False),
                       ("sort min(total,31)",lambda i:np.minimum((lo+hi)[i],31),False)]:
    t=run(kf,single); print("%-28s iterations %d -> %.1f VALU/block (30/iter)"%(name,t,t*30/nblocks))
print("---- keys")
for name,kf in [("hi*64+lo",lambda i:hi[i]*64+lo[i]),("lo*64+hi",lambda i:lo[i]*64+hi[i]),("lo+2hi",lambda i:lo[i]+2*hi[i]),("lo+4hi",lambda i:lo[i]+4*hi[i]),("max(lo,4hi)",lambda i:np.maximum(lo[i],4*hi[i]))]:
    t=run(kf,False); print("%-28s iterations %d -> %.1f VALU/block (30/iter)"%(name,t,t*30/nblocks))
print("---- team sizes (blocks per wave-pass = 64/lanes_per_team)")
def run2(bpw, keyfun, order_mode):
    tot=0
    for my in range(mcuy):
        for gx in range(mcux//G):
            b0=(my*mcux+gx*G)*6
            idx=np.arange(b0,b0+96)
            if order_mode=='sorted': order=idx[np.argsort(-keyfun(idx),kind='stable')]
            else:
                ss=np.arange(96); order=np.where(ss<64,(ss>>2)*6+(ss&3),((ss-64)>>1)*6+4+((ss-64)&1))+b0
            for s0 in range(0,96,bpw):
                bl=order[s0:s0+bpw]
                tot+= lo[bl].max()+hi[bl].max()
    return tot
for lanes,core in [(4,16),(2,32),(1,64)]:
    bpw=64//lanes
    for mode in ('compmajor','sorted'):
        t=run2(bpw,lambda i:np.maximum(lo[i],4*hi[i]),mode)
        print("lanes/team %d %-10s iterations/group-pass total %d -> VALU/block %.1f (bookkeeping 14 + %d pk)"%(lanes,mode,t,t*(14+core)/nblocks,core))
