#!/usr/bin/env python3
"""CPU study behind the choice of the f16x3 arithmetic: a numpy restatement of
the network in which every matrix product can be evaluated as
    f32 / f64        plain
    h1               fp16 operands
    h3 / h3z / h4    fp16 split, 3 products (subnormals kept / flushed), 4 products
    b3 / b6 / b9     bf16 split, 3 products (2 parts) / 6 products (3 parts) / all 9 products of 3 parts (exact operands:
                     three bf16 pieces carry all 24 bits of an fp32 value)
with fp32 results, run over the 60 s (16 kHz) and 169 s (8 kHz) speech fixtures and compared with the
golden probabilities recorded from the reference model.
    python tests/study_split_precision.py f32 h1 h3 h3z b3 b6
"""
import sys, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from oracle.weights import read_container
from oracle import Oracle
W = read_container(open(str(__import__('pathlib').Path(__file__).resolve().parents[1] / 'silero_vad_amd/data/silero_vad_v6.weights'),'rb').read())
f32=np.float32
def split16(x, n=2, dt=np.float16, ftz=False):
    parts=[]; r=x.astype(np.float64)
    for i in range(n):
        h=r.astype(f32).astype(dt)
        if ftz and dt==np.float16:
            h=np.where(np.abs(h.astype(f32))<6.1035e-5, 0, h).astype(dt)
        parts.append(h.astype(np.float64)); r=r-parts[-1]
    return parts
def bf16(x):
    u=x.astype(f32).view(np.uint32).astype(np.uint64)
    u=((u+0x7fff+((u>>16)&1))>>16)<<16
    return u.astype(np.uint32).view(f32)
def splitbf(x,n):
    parts=[]; r=x.astype(np.float64)
    for i in range(n):
        h=bf16(r.astype(f32)).astype(np.float64); parts.append(h); r=r-h
    return parts
MODE='f32'
def mm(A,Bm):
    # A [M,K] weights, Bm [K,N] activations; returns f32
    if MODE=='f32': return (A.astype(f32)@Bm.astype(f32))
    if MODE=='f64': return (A.astype(np.float64)@Bm.astype(np.float64)).astype(f32)
    if MODE.startswith('h'):  # h3, h3z, h1
        ftz = MODE.endswith('z'); n=int(MODE[1])
        a=split16(A,2,ftz=ftz); b=split16(Bm,2,ftz=ftz)
        if n==1: return (a[0]@b[0]).astype(f32)
        if n==3: return (a[0]@b[0]+a[0]@b[1]+a[1]@b[0]).astype(f32)
        if n==4: return (a[0]@b[0]+a[0]@b[1]+a[1]@b[0]+a[1]@b[1]).astype(f32)
    if MODE.startswith('b'):
        n=int(MODE[1])
        if n==3:
            a=splitbf(A,2); b=splitbf(Bm,2); return (a[0]@b[0]+a[0]@b[1]+a[1]@b[0]).astype(f32)
        if n==6:
            a=splitbf(A,3); b=splitbf(Bm,3)
            return (a[0]@b[0]+a[0]@b[1]+a[1]@b[0]+a[1]@b[1]+a[0]@b[2]+a[2]@b[0]).astype(f32)
        if n==9:
            a=splitbf(A,3); b=splitbf(Bm,3)
            acc=np.zeros((A.shape[0],Bm.shape[1]),f32)
            for i in range(3):
                for j in range(3):
                    acc=(acc+(a[i]@b[j]).astype(f32)).astype(f32)          # nine fp32 partial sums, added in fp32
            return acc
    raise ValueError(MODE)
def sigmoid(x): return (1/(1+np.exp(-x.astype(np.float64)))).astype(f32)
class Net:
    def __init__(s,sr):
        p='_model' if sr==16000 else '_model_8k'
        s.N=512 if sr==16000 else 256; s.C=s.N//8; s.F=s.N//2; s.H=s.F//2; s.K=s.F//2+1
        s.basis=W[p+'.stft.forward_basis_buffer'].reshape(2*s.K,s.F).astype(np.float64)
        s.ew=[W[p+f'.encoder.{l}.reparam_conv.weight'] for l in range(4)]
        s.eb=[W[p+f'.encoder.{l}.reparam_conv.bias'] for l in range(4)]
        s.wih=W[p+'.decoder.rnn.weight_ih']; s.whh=W[p+'.decoder.rnn.weight_hh']
        s.b=(W[p+'.decoder.rnn.bias_ih']+W[p+'.decoder.rnn.bias_hh'])
        s.wo=W[p+'.decoder.decoder.2.weight'].reshape(128); s.bo=W[p+'.decoder.decoder.2.bias'].reshape(())
    def front(s,x1):  # x1 [B,C+N] -> enc3 [B,128]
        B=x1.shape[0]; C,N,F,H,K=s.C,s.N,s.F,s.H,s.K
        xp=np.concatenate([x1, x1[:, C+N-2:C+N-2-C:-1]],1)
        fr=np.stack([xp[:,m*H:m*H+F] for m in range(4)],1).astype(np.float64)  # B,4,F
        sp=fr@s.basis.T  # B,4,2K
        mag=np.sqrt(sp[...,:K]**2+sp[...,K:]**2).astype(f32)  # B,4,K
        X=mag.transpose(0,2,1)  # B,K,4
        strides=[1,2,2,1]
        for l in range(4):
            w=s.ew[l]; Co,Ci,_=w.shape; T=X.shape[2]; st=strides[l]; To=(T+2-3)//st+1
            Xp=np.zeros((B,Ci,T+2),f32); Xp[:,:,1:T+1]=X
            Y=np.zeros((B,Co,To),f32)
            for u in range(To):
                col=Xp[:,:,u*st:u*st+3]  # B,Ci,3
                # GEMM: w [Co, Ci*3] x col [Ci*3, B]
                Y[:,:,u]=mm(w.reshape(Co,Ci*3), col.reshape(B,Ci*3).T).T + s.eb[l]
            X=np.maximum(Y,0)
        return X[:,:,0]
    def run(s,pcm,T=None):
        N,C=s.N,s.C; pcm=np.atleast_2d(pcm).astype(f32); B=pcm.shape[0]
        T=T or pcm.shape[1]//N
        ctx=np.zeros((B,C),f32); h=np.zeros((B,128),f32); c=np.zeros((B,128),f32)
        # frontend for all chunks at once
        feats=[]
        x=np.concatenate([ctx,pcm[:,:T*N]],1)
        X1=np.stack([x[:,t*N:t*N+N+C] for t in range(T)],1).reshape(B*T,N+C)
        fe=s.front(X1).reshape(B,T,128)
        gx=mm(s.wih, fe.reshape(B*T,128).T).T.reshape(B,T,512)+s.b
        probs=np.zeros((B,T),f32)
        for t in range(T):
            g=gx[:,t]+mm(s.whh,h.T).T
            i=sigmoid(g[:,:128]); f=sigmoid(g[:,128:256]); gg=np.tanh(g[:,256:384].astype(np.float64)).astype(f32); o=sigmoid(g[:,384:])
            c=f*c+i*gg; h=o*np.tanh(c.astype(np.float64)).astype(f32)
            probs[:,t]=sigmoid(np.maximum(h,0)@s.wo+s.bo)
        return probs,h,c
if __name__=='__main__':
    S = sys.modules[__name__]
    for sr,fa,fg in ((16000,'audio_16k','golden_16k'),(8000,'audio_8k','golden_8k')):
        pcm=np.load(str(__import__('pathlib').Path(__file__).resolve().parents[1] / f'tests/golden/{fa}.npz'))['pcm'].astype(f32)/32768
        G=np.load(str(__import__('pathlib').Path(__file__).resolve().parents[1] / f'tests/golden/{fg}.npz'))
        net=Net(sr)
        for mode in sys.argv[1:]:
            S.MODE=mode
            globals()['MODE']=mode
            p,h,c=net.run(pcm)
            gp=G['probs_wav']; n=min(len(gp),p.shape[1])
            d=np.abs(p[0,:n]-gp[:n])
            print(sr,mode,'max|dp|=%.3e mean=%.3e'%(d.max(),d.mean()),'dh=%.3e dc=%.3e'%(np.abs(h-G['state_wav'][0]).max(),np.abs(c-G['state_wav'][1]).max()))
