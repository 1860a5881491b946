"""Experiment (CPU, pure Python): how many bits does a DEFLATE token decoder started at a wrong bit position need to fall in
with the true token chain?  Streams of the oracle encoder (default effort, 256 KiB strips) on the bench text; 400 random
starts per segment.  k_inflate_tokens_spec's look-back (512 bits) rests on the percentiles this prints."""
import sys, numpy as np, random
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import oracle_lib as O
from starflate_amd import synth

LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

class BR:
    def __init__(s, data, pos=0): s.d=data; s.p=pos
    def bit(s):
        b=(s.d[s.p>>3]>>(s.p&7))&1; s.p+=1; return b
    def bits(s,n):
        v=0
        for i in range(n): v|=s.bit()<<i
        return v

def mk(lens):
    # canonical code -> dict (len, code) -> sym
    maxl=max(lens) if len(lens) else 0
    cnt=[0]*(maxl+2)
    for l in lens: cnt[l]+=1
    cnt[0]=0; code=0; nxt=[0]*(maxl+2)
    for l in range(1,maxl+1):
        code=(code+cnt[l-1])<<1; nxt[l]=code
    t={}
    for s,l in enumerate(lens):
        if l: t[(l,nxt[l])]=s; nxt[l]+=1
    return t,maxl

def dec(br,t,maxl):
    c=0
    for l in range(1,maxl+1):
        c=(c<<1)|br.bit()
        if (l,c) in t: return t[(l,c)]
    return None

def header(br):
    fin=br.bits(1); ty=br.bits(2)
    assert ty==2, ty
    hlit=br.bits(5)+257; hdist=br.bits(5)+1; hclen=br.bits(4)+4
    order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
    cl=[0]*19
    for i in range(hclen): cl[order[i]]=br.bits(3)
    ct,cm=mk(cl)
    lens=[]
    while len(lens)<hlit+hdist:
        s=dec(br,ct,cm)
        if s<16: lens.append(s)
        elif s==16: lens+= [lens[-1]]*(3+br.bits(2))
        elif s==17: lens+=[0]*(3+br.bits(3))
        else: lens+=[0]*(11+br.bits(7))
    return mk(lens[:hlit]), mk(lens[hlit:])

def token(br,L,D,end):
    """one token from br.p; returns False on error / eob"""
    try:
        s=dec(br,*L)
        if s is None or s>285: return False
        if s<256: return True
        if s==256: return False
        br.bits(LEXT[s-257])
        d=dec(br,*D)
        if d is None or d>29: return False
        br.bits(DEXT[d])
        return br.p<=end
    except IndexError:
        return False

def run(kind, nseg=6, trials=400):
    data = synth.gen_text(nseg*32768, seed=3) if kind=='text' else None
    if kind!='text':
        from starflate_amd import realbytes
        data = realbytes.load(kind, nseg*32768+ (1<<20))[(1<<20):(1<<20)+nseg*32768]
    P=O.default_params(strip_bytes=262144)
    out=O.compress(data,P)
    idx=O.last_index() if hasattr(O,'last_index') else None
    return data,out

if __name__=='__main__':
    kind=sys.argv[1] if len(sys.argv)>1 else 'text'
    data,out=run(kind)
    d=bytes(out)
    # walk segments: each = dynamic block + empty stored block
    pos=0; res=[]; random.seed(1); fails=0; nseg=0
    while pos<len(d) and nseg<6:
        br=BR(d,pos*8)
        L,D=header(br)
        start=br.p
        bounds=[start]
        while True:
            p0=br.p
            s=dec(br,*L)
            if s==256: break
            if s>=257:
                br.bits(LEXT[s-257]); dd=dec(br,*D); br.bits(DEXT[dd])
            bounds.append(br.p)
        eobpos=p0; end=br.p
        bset=set(bounds)
        # stored empty block: 3 bits, align, 4 bytes
        p=br.p+3; p=(p+7)//8*8; p+=32; pos=p//8; nseg+=1
        for _ in range(400):
            g=random.randrange(start, max(start+1,eobpos-3000))
            if g in bset: continue
            b2=BR(d,g); ok=True
            while b2.p not in bset:
                if not token(b2,L,D,eobpos): ok=False; break
            if ok: res.append(b2.p-g)
            else: fails+=1
        print('segment', nseg, 'body bits', eobpos-start, 'tokens', len(bounds)-1, flush=True)
    r=np.array(res)
    print(kind, 'trials', len(r), 'fails(err before sync)', fails)
    for q in (50,90,99,99.9,100): print(q, np.percentile(r,q))
