"""CPU executor of the op program (include/iss.h ISS_OP_* rows) -- test infrastructure.

Runs what inaspeechsegmenter_amd/keras_model.py lowers a layer list to, with torch-CPU float64 convolutions, so that the
lowering itself (BatchNorm folding, fused activations / pools, channel padding, flatten maps, buffer ping-pong) can be
checked against the Keras-semantics oracle without a GPU.  Input: z-normalised patches (N,H,W,C) for a patch network.
"""
import numpy as np

from inaspeechsegmenter_amd import _native as N


def run(comp, x):
    import torch
    import torch.nn.functional as F
    prog, blob = np.asarray(comp.prog), np.asarray(comp.blob, dtype=np.float64)
    bufs = {N.BUF_INPUT: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))}
    out = None
    for R in prog:
        src = bufs[int(R[N.C_IN])]
        op = int(R[N.C_OP])
        h, w, cin, ho, wo, cout = (int(R[c]) for c in (N.C_H, N.C_W, N.C_CIN, N.C_HO, N.C_WO, N.C_COUT))
        src = src.reshape(-1, h, w, cin)
        if op == N.OP_CONV:
            kh, kw, sh, sw, pt, pl = (int(R[c]) for c in (N.C_KH, N.C_KW, N.C_SH, N.C_SW, N.C_PT, N.C_PL))
            K = kh * kw * cin
            kpad = -(-K // N.K_ALIGN) * N.K_ALIGN
            W = blob[R[N.C_WOFF]:R[N.C_WOFF] + cout * kpad].reshape(cout, kpad)[:, :K].reshape(cout, kh, kw, cin)
            t = src.permute(0, 3, 1, 2)
            pb = max((ho - 1) * sh + kh - h - pt, 0)
            pr = max((wo - 1) * sw + kw - w - pl, 0)
            t = F.pad(t, (pl, pr, pt, pb))
            y = F.conv2d(t, torch.from_numpy(np.ascontiguousarray(W.transpose(0, 3, 1, 2))), stride=(sh, sw))[:, :, :ho, :wo]
            y = y.permute(0, 2, 3, 1)
            if R[N.C_BOFF] >= 0:
                y = y + torch.from_numpy(blob[R[N.C_BOFF]:R[N.C_BOFF] + cout])
            if R[N.C_RES] >= 0:
                y = y + bufs[int(R[N.C_RES])].reshape(y.shape)
            act = int(R[N.C_ACT])
            y = [y, torch.relu(y), torch.sigmoid(y), torch.tanh(y)][act]
            if R[N.C_PSOFF] >= 0:
                y = y * torch.from_numpy(blob[R[N.C_PSOFF]:R[N.C_PSOFF] + cout]) + torch.from_numpy(blob[R[N.C_PTOFF]:R[N.C_PTOFF] + cout])
            ph, pw = max(int(R[N.C_FPOOLH]), 1), max(int(R[N.C_FPOOLW]), 1)
            if ph * pw > 1:
                y = y[:, :ho // ph * ph, :wo // pw * pw].reshape(-1, ho // ph, ph, wo // pw, pw, cout)
                y = y.amax(dim=(2, 4)) if R[N.C_POOLKIND] == 0 else y.mean(dim=(2, 4))
            out = y
        elif op == N.OP_POOL:
            kh, kw, sh, sw, pt, pl = (int(R[c]) for c in (N.C_KH, N.C_KW, N.C_SH, N.C_SW, N.C_PT, N.C_PL))
            t = src.permute(0, 3, 1, 2)
            pb = max((ho - 1) * sh + kh - h - pt, 0)
            pr = max((wo - 1) * sw + kw - w - pl, 0)
            if R[N.C_POOLKIND] == 0:
                t = F.pad(t, (pl, pr, pt, pb), value=float('-inf'))
                y = F.max_pool2d(t, (kh, kw), (sh, sw))
            else:                                        # mean over the elements inside the input (padded 'same' average pools)
                ones = F.pad(torch.ones_like(t[:1, :1]), (pl, pr, pt, pb))
                y = F.avg_pool2d(F.pad(t, (pl, pr, pt, pb)), (kh, kw), (sh, sw)) / F.avg_pool2d(ones, (kh, kw), (sh, sw))
            out = y[:, :, :ho, :wo].permute(0, 2, 3, 1)
        elif op == N.OP_SOFTMAX:
            out = torch.softmax(src, dim=-1)
        elif op == N.OP_ACT:                         # elementwise elu / leaky relu / selu / softplus (in place in the program)
            alpha = float(np.array([int(R[N.C_ACTPARAM])], np.int32).view(np.float32)[0])
            out = {4: lambda: F.elu(src, alpha=alpha), 5: lambda: F.leaky_relu(src, negative_slope=alpha), 6: lambda: F.selu(src),
                   7: lambda: F.softplus(src), 8: lambda: torch.clamp(src, 0.0, alpha),
                   9: lambda: (lambda mv, th: torch.where(src > th, torch.clamp(src, max=mv), alpha * (src - th)))(
                       float(np.array([int(R[N.C_ACTPARAM2])], np.int32).view(np.float32)[0]),
                       float(np.array([int(R[N.C_ACTPARAM3])], np.int32).view(np.float32)[0]))}[int(R[N.C_ACT])]()
        elif op == N.OP_ELT:                         # merge / data-movement rows of graph-shaped models
            kind = int(R[N.C_ACT])
            if N.ELT_ADD <= kind <= N.ELT_AVG:
                b = bufs[int(R[N.C_RES])].reshape(src.shape)
                out = {N.ELT_ADD: lambda: src + b, N.ELT_SUB: lambda: src - b, N.ELT_MUL: lambda: src * b,
                       N.ELT_MAX: lambda: torch.maximum(src, b), N.ELT_MIN: lambda: torch.minimum(src, b),
                       N.ELT_AVG: lambda: (src + b) * 0.5}[kind]()
                if int(R[N.C_ORDER]) == 1:
                    out = torch.relu(out)
            elif kind in (N.ELT_COPY, N.ELT_ZERO):
                nch, soff, doff = int(R[N.C_KH]), int(R[N.C_PT]), int(R[N.C_PL])
                old = bufs.get(int(R[N.C_OUT]))
                if kind == N.ELT_ZERO:
                    out = src.reshape(-1, h, w, cout).clone()
                    out[..., doff:doff + nch] = 0
                else:
                    out = old.reshape(-1, h, w, cout).clone() if old is not None and old.numel() == len(src) * h * w * cout \
                        else torch.full((len(src), h, w, cout), float('nan'), dtype=torch.float64)   # NaN: channels nobody wrote show up
                    out[..., doff:doff + nch] = src[..., soff:soff + nch]
            elif kind == N.ELT_PERMUTE:
                out = src.permute(0, 1 + int(R[N.C_KH]), 1 + int(R[N.C_KW]), 1 + int(R[N.C_SH]))
                assert tuple(out.shape[1:]) == (ho, wo, cout)
            else:
                raise NotImplementedError(kind)
        elif op == N.OP_STATPOOL:                    # mean || std over W (time) per (h, c), torch's (c, h) flatten order
            mean = src.mean(dim=2)                   # (N, H, C)
            std = torch.sqrt((src * src).mean(dim=2) - mean * mean + 1e-10)
            out = torch.cat((mean.permute(0, 2, 1).reshape(len(src), -1), std.permute(0, 2, 1).reshape(len(src), -1)), 1)
        else:
            raise NotImplementedError(op)
        bufs[int(R[N.C_OUT])] = out.contiguous()
    return out.reshape(len(x), -1).numpy()
