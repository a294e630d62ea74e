#!/usr/bin/env python3
"""CPU study behind the choice of the Winograd forms of encoder 0: a numpy float32 restatement of
the frontend in which encoder 0 is evaluated tap by tap (`direct`), as two F(2,3) tiles (`w2`) or as one F(4,3) tile
with interpolation points (0, +-1, +-2, inf) (`w4`, the product's form), (0, +-1, +-1/2, inf) (`w4h`) or mixed (`w4m`),
compared with a float64 evaluation (encoder-0 output, probabilities) and with the golden probabilities recorded from the
reference model -- on real speech, loud noise, speech at 1e-3 and un-normalised int16-range input.
    python tests/study_winograd_numerics.py
"""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import study_split_precision as S
f32 = np.float32
MODE = 'direct'
BT = {2: None, 4: np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], float)}
G4 = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
AT4 = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], float)
# alternative points (0, +-1, +-1/2, inf)
def cook_toom(points):
    # F(4,3): returns AT (4x6), G (6x3), BT (6x6) via Vandermonde construction, polynomial points + infinity
    import numpy.polynomial.polynomial as P
    n = 6; m = 4; r = 3
    pts = list(points)  # 5 finite points
    AT = np.zeros((m, n)); Gm = np.zeros((n, r)); 
    for i, p in enumerate(pts):
        AT[:, i] = [p**k for k in range(m)]
        Gm[i, :] = [p**k for k in range(r)]
    AT[m-1, n-1] = 1; Gm[n-1, r-1] = 1
    # BT from requirement: y = AT [(G g) * (BT d)] equals correlation -> solve by linear algebra
    # unknown BT (6x6): for all g, d: sum_i AT[o,i] (G g)_i (BT d)_i = sum_k g_k d_{o+k}
    # => for each o,k,j: sum_i AT[o,i] G[i,k] BT[i,j] = [j == o+k]
    rows = []; rhs = []
    for o in range(m):
        for k in range(r):
            rows.append(AT[o, :] * Gm[:, k]); 
    M = np.array(rows)  # 12 x 6
    R = np.zeros((12, 6))
    idx = 0
    for o in range(m):
        for k in range(r):
            R[idx, o + k] = 1; idx += 1
    BTm, res, rk, sv = np.linalg.lstsq(M, R, rcond=None)
    assert np.abs(M @ BTm - R).max() < 1e-9, np.abs(M @ BTm - R).max()
    return AT, Gm, BTm
def enc0(net, X):   # X [B, K, 4] f32 -> Y [B,128,4]
    w = net.ew[0].astype(np.float64); b = net.eb[0]; B = X.shape[0]
    if MODE == 'direct' or MODE == 'f64':
        Xp = np.zeros((B, X.shape[1], 6), f32); Xp[:, :, 1:5] = X
        Y = np.zeros((B, 128, 4), f32)
        for u in range(4):
            col = Xp[:, :, u:u+3].reshape(B, -1).T
            if MODE == 'f64': Y[:, :, u] = (w.reshape(128, -1) @ col.astype(np.float64)).T.astype(f32) + b
            else: Y[:, :, u] = (w.reshape(128, -1).astype(f32) @ col).T + b
        return Y
    if MODE == 'w2':
        GA = ((w[:, :, 0] + w[:, :, 1] + w[:, :, 2]) / 2).astype(f32); GB = ((w[:, :, 0] - w[:, :, 1] + w[:, :, 2]) / 2).astype(f32)
        G0 = w[:, :, 0].astype(f32); G2 = w[:, :, 2].astype(f32)
        Y = np.zeros((B, 128, 4), f32)
        Xp = np.zeros((B, X.shape[1], 6), f32); Xp[:, :, 1:5] = X
        for pr in range(2):
            d = [Xp[:, :, 2 * pr + i] for i in range(4)]
            m0 = G0 @ (d[0] - d[2]).T; m1 = GA @ (d[1] + d[2]).T; m2 = GB @ (d[2] - d[1]).T; m3 = G2 @ (d[1] - d[3]).T
            Y[:, :, 2 * pr] = (m0 + (m1 + m2)).T + b; Y[:, :, 2 * pr + 1] = ((m1 - m2) - m3).T + b
        return Y
    if MODE.startswith('w4'):
        AT, Gm, BTm = WM
        Xp = np.zeros((B, X.shape[1], 6), f32); Xp[:, :, 1:5] = X
        t = [sum(f32(BTm[i, j]) * Xp[:, :, j] for j in range(6) if BTm[i, j] != 0).astype(f32) for i in range(6)]
        U = [(w @ Gm[i]).astype(f32) for i in range(6)]       # [128, K]
        m = [U[i] @ t[i].T for i in range(6)]
        Y = np.zeros((B, 128, 4), f32)
        for o in range(4):
            Y[:, :, o] = sum(f32(AT[o, i]) * m[i] for i in range(6) if AT[o, i] != 0).astype(f32).T + b
        return Y
orig_front = S.Net.front
def front(s, x1):
    B=x1.shape[0]; C,N,F,H,K=s.C,s.N,s.F,s.H,s.K
    xp=np.concatenate([x1, x1[:, C+N-2:C+N-2-C:-1]],1)
    fr=np.stack([xp[:,m*H:m*H+F] for m in range(4)],1).astype(np.float64)
    sp=fr@s.basis.T
    mag=np.sqrt(sp[...,:K]**2+sp[...,K:]**2).astype(f32)
    X=mag.transpose(0,2,1)
    Y = enc0(s, X); 
    s.last_e0 = Y
    X = np.maximum(Y, 0)
    strides=[1,2,2,1]
    for l in range(1,4):
        w=s.ew[l]; Co,Ci,_=w.shape; T=X.shape[2]; st=strides[l]; To=(T+2-3)//st+1
        Xp=np.zeros((B,Ci,T+2),f32); Xp[:,:,1:T+1]=X
        Y=np.zeros((B,Co,To),f32)
        for u in range(To):
            col=Xp[:,:,u*st:u*st+3]
            Y[:,:,u]=S.mm(w.reshape(Co,Ci*3), col.reshape(B,Ci*3).T).T + s.eb[l]
        X=np.maximum(Y,0)
    return X[:,:,0]
S.Net.front = front
S.MODE = 'f32'
root = str(ROOT) + '/'
for sr, fa, fg in ((16000,'audio_16k','golden_16k'),):
    pcm = np.load(root + f'tests/golden/{fa}.npz')['pcm'].astype(f32) / 32768
    G = np.load(root + f'tests/golden/{fg}.npz')
    net = S.Net(sr)
    rng = np.random.default_rng(0)
    cases = {'speech': pcm, 'loud_noise': rng.standard_normal(16000 * 20).astype(f32).clip(-1, 1), 'speech_x1e-3': pcm * f32(1e-3),
             'unnorm_int16': pcm * f32(32768)}
    for cname, x in cases.items():
        MODE = 'f64'; pref, _, _ = net.run(x); e_ref = net.last_e0.astype(np.float64)
        for mode, pts in (('direct', None), ('w2', None), ('w4', (0, 1, -1, 2, -2)), ('w4h', (0, 1, -1, 0.5, -0.5)), ('w4m', (0, 1, -1, 2, -0.5))):
            MODE = mode
            if pts: WM = cook_toom(pts)
            p, h, c = net.run(x)
            e = net.last_e0
            line = f'{cname:14s} {mode:7s} enc0 rel err {np.abs(e - e_ref).max() / np.abs(e_ref).max():.2e}  dp vs f64 {np.abs(p - pref).max():.2e}'
            if cname == 'speech':
                gp = G['probs_wav']; n = min(len(gp), p.shape[1]); line += f'  dp vs golden {np.abs(p[0, :n] - gp[:n]).max():.2e}'
            print(line, flush=True)
