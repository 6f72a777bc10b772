"""Plain-PyTorch fp32 references of every op in gen6d_amd/ops.py (same signatures, same in-place output contract).

Test infrastructure only.  Two uses:
  * `-m gpu` tests compare each HIP kernel with its reference here on the same inputs;
  * `-m "not gpu"` tests monkeypatch gen6d_amd.ops with this module (see `patch_ops`) so the HOST orchestration of
    gen6d_amd/network/* (layouts, weight repacking, InstanceNorm fusion, buffer slicing) is validated against the CPU
    oracle without a GPU.  The product never imports this file.
"""
import torch
import torch.nn.functional as F


def workspace(device):
    return None


def fork_join(fns, device):
    return [fn() for fn in fns]


def _act(v, act):
    if act == 1: return F.relu(v)
    if act == 2: return F.leaky_relu(v, 0.1)
    return v


def conv(x, w, bias, out, ksize=(1, 1, 1), stride=(1, 1, 1), pad=(0, 0, 0), mul=None, in_scale=None, in_shift=None,
         in_relu=False, per_n=False, out_act=0, stats=None, rows_per_group=0, split_k=0, w_wino=None, finalize=None, eps=1e-5, w_wino43=None,
         in_mod=0, mul_group=0):
    N = out.shape[0] if in_mod else x.shape[0]
    _, Di, Hi, Wi, Cin = x.shape
    Cout = w.shape[0]
    v = x
    if in_mod: v = x[torch.arange(N) % in_mod]                      # the queries of a batch share the input images
    if mul is not None:
        v = v * (mul[torch.arange(N) // mul_group][:, None] if mul_group else mul[None, None])
    per_n = int(per_n)
    if in_scale is not None:
        if per_n:
            g = torch.arange(N) // per_n                               # a table per run of per_n images
            v = v * in_scale.reshape(-1, Cin)[g].view(N, 1, 1, 1, Cin) + in_shift.reshape(-1, Cin)[g].view(N, 1, 1, 1, Cin)
        else: v = v * in_scale.view(1, 1, 1, 1, Cin) + in_shift.view(1, 1, 1, 1, Cin)
    if in_relu: v = F.relu(v)
    w5 = w.view(Cout, ksize[0], ksize[1], ksize[2], Cin).permute(0, 4, 1, 2, 3)
    y = F.conv3d(v.permute(0, 4, 1, 2, 3), w5, bias, stride=stride, padding=pad).permute(0, 2, 3, 4, 1)
    y = _act(y, out_act)
    out.copy_(y)
    if stats is not None:
        M = y.numel() // Cout
        rpg = rows_per_group if rows_per_group > 0 else M
        g = y.reshape(M // rpg, rpg, Cout).double()
        stats[:, :, 0] += g.sum(1)
        stats[:, :, 1] += (g * g).sum(1)
    if finalize is not None:
        return stats_finalize(stats, finalize, eps)
    return out


def corr2d_patch(x, w, out, k):
    return conv(x, w, None, out, ksize=(1, k, k), pad=(0, k // 2, k // 2))


def corr2d_patch_multi(xs, w, outs, k):
    for x, o in zip(xs, outs):                                     # [N,1,H,W,C]: the conv reference handles the batch axis
        corr2d_patch(x, w, o, k)
    return outs


def corr2d_wino_multi(xs, U, outs, kblocks=5):
    """The block-wise Winograd ALGORITHM on the transformed filters (not F.conv2d): out = sum over the kblocks^2 blocks of the
    3x3 Winograd convolution (wino_conv3x3 above) of the input shifted by (3bi - 3(kb-1)/2, 3bj - ...) — checks the host-side
    block cut / transform / ordering (backbone.winograd_corr_filters) together with the algebra the kernel implements."""
    kb = kblocks
    for x, o in zip(xs, outs):
        N, _, H, W, Cin = x.shape
        nc = Cin // 8
        acc = torch.zeros((N, H, W, U.shape[2]), dtype=x.dtype)
        pad = 3 * (kb - 1) // 2 + 1                                     # + the 3x3 halo of every shifted window
        xp = F.pad(x[:, 0], (0, 0, pad, pad, pad, pad))
        zero_b = torch.zeros(U.shape[2], dtype=x.dtype)
        for bi in range(kb):
            for bj in range(kb):
                b = bi * kb + bj
                sh = xp[:, 3 * bi:3 * bi + H + 2, 3 * bj:3 * bj + W + 2]  # input shifted by (3bi - 6, 3bj - 6) with its halo, zero outside the image
                y, _ = wino_conv3x3(sh.contiguous(), U.view(nc, kb * kb, *U.shape[1:])[:, b].contiguous(), zero_b, relu=False)   # chunk-major rows
                acc += y[:, 1:H + 1, 1:W + 1]
        o.copy_(acc[:, None])
    return outs


def stats_arena_begin(device):
    pass


def new_stats(groups, channels, device):
    return torch.zeros((groups, channels, 2), dtype=torch.float64, device=device)


def stats_finalize(stats, count, eps=1e-5):
    mean = stats[..., 0] / count
    var = (stats[..., 1] / count - mean * mean).clamp_min(0)
    rs = 1.0 / torch.sqrt(var + eps)
    return rs.float(), (-mean * rs).float()


def _aff(x, scale, shift, per_n, relu):
    N, C = x.shape[0], x.shape[-1]
    if scale is not None:
        if per_n:
            g = torch.arange(N) // int(per_n)
            x = x * scale.reshape(-1, C)[g].view(N, 1, 1, 1, C) + shift.reshape(-1, C)[g].view(N, 1, 1, 1, C)
        else:
            x = x * scale.reshape(1, 1, 1, 1, C) + shift.reshape(1, 1, 1, 1, C)
    return F.relu(x) if relu else x


def affine_act_pool(x, out, scale=None, shift=None, per_n=False, relu=False, pool=0):
    N, D, H, W, C = x.shape
    if per_n and D != 1: raise ValueError
    v = _aff(x, scale, shift, per_n, relu)
    if pool == 1:
        v = F.max_pool3d(v.permute(0, 4, 1, 2, 3), (1, 2, 2), (1, 2, 2)).permute(0, 2, 3, 4, 1)
    elif pool == 2:
        v = v.mean((2, 3), keepdim=True)
    out.copy_(v)
    return out


def upsample_bilinear(x, out, factor, scale=None, shift=None, per_n=False):
    N, D, H, W, C = x.shape
    v = _aff(x, scale, shift, per_n, False).reshape(N * D, H, W, C).permute(0, 3, 1, 2)
    v = F.interpolate(v, scale_factor=factor, mode="bilinear", align_corners=False)
    out.copy_(v.permute(0, 2, 3, 1).reshape(N, D, H * factor, W * factor, C))
    return out


def bias_relu_pool_nchw(x, bias, relu, pool):
    v = x + bias.view(1, -1, 1, 1)
    if relu: v = F.relu(v)
    return F.max_pool2d(v, 2, 2) if pool else v


def vgg_conv1_pool(x, w_oihw, bias):
    return F.max_pool2d(F.relu(F.conv2d(x, w_oihw, bias, padding=1)), 2, 2)


def vgg_conv1_pool_nhwc(x, w_oihw, bias, out=None, norm=None):
    if norm is not None:
        x = (x - torch.tensor(norm[0], dtype=x.dtype).view(1, 3, 1, 1)) / torch.tensor(norm[1], dtype=x.dtype).view(1, 3, 1, 1)
    y = vgg_conv1_pool(x, w_oihw, bias).permute(0, 2, 3, 1).contiguous()
    if out is None:
        return y
    out.copy_(y)
    return out


def alloc_like_segments(shapes, device):
    return [torch.empty(sh, dtype=torch.float32, device=device) for sh in shapes]


def wino_conv3x3_multi(xs, U, bias, relu=True, full=True, pool=False):
    res = [wino_conv3x3(x, U, bias, relu, full, pool) for x in xs]
    return ([r[0] for r in res] if full else None), ([r[1] for r in res] if pool else None)


def wino_conv3x3(x, U, bias, relu=True, full=True, pool=False):
    """The Winograd F(2x2,3x3) algorithm itself (not F.conv2d), from the TRANSFORMED filters the kernel receives: checks
    the host-side filter transform / layout (backbone.winograd_filters) together with the algebra the kernel implements."""
    N, H, W, Cin = x.shape
    Cout = U.shape[2]
    dt = x.dtype
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=dt)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=dt)
    Ht, Wt = (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x, (0, 0, 1, 2 * Wt + 1 - W, 1, 2 * Ht + 1 - H))                       # [N, 2Ht+2, 2Wt+2, C]
    d = xp.unfold(1, 4, 2).unfold(2, 4, 2)                                             # [N,Ht,Wt,C,4,4]
    V = torch.einsum("ai,ntucij,bj->ntuabc", BT, d, BT)
    U = U.clone()
    swap = (torch.arange(Cout) & 8) != 0                                               # undo the LDS-bank swizzle of the halves
    U[:, :, swap] = torch.cat([U[:, :, swap, 4:], U[:, :, swap, :4]], -1)
    U4 = U.permute(1, 2, 0, 3).reshape(4, 4, Cout, Cin).to(dt)
    M = torch.einsum("ntuabc,aboc->ntuabo", V, U4)
    Y = torch.einsum("pa,ntuabo,qb->ntupqo", AT, M, AT)                               # [N,Ht,Wt,2,2,Cout]
    y = Y.permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * Ht, 2 * Wt, Cout)[:, :H, :W] + bias.to(dt)
    if relu:
        y = F.relu(y)
    yp = F.max_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous() if pool else None
    return (y.contiguous() if full else None), yp


def _wino43(x, U43, bias, relu, full, pool):
    """The F(4x4,3x3) ALGORITHM of csrc/wino43_conv.hip on the transformed filters the kernel receives (layout of
    backbone.winograd43_filters), in the dtype of x."""
    from gen6d_amd.network.backbone import winograd43_matrices
    N, H, W, Cin = x.shape
    Cout = U43.shape[2] * U43.shape[4] * 32
    dt = x.dtype
    BT, _, AT = winograd43_matrices(dtype=dt)
    Ht, Wt = (H + 3) // 4, (W + 3) // 4
    xp = F.pad(x, (0, 0, 1, 4 * Wt + 1 - W, 1, 4 * Ht + 1 - H))
    d = xp.unfold(1, 6, 4).unfold(2, 6, 4)                                             # [N,Ht,Wt,C,6,6]
    V = torch.einsum("ai,ntucij,bj->ntuabc", BT, d, BT)
    nblk, npp = U43.shape[2], U43.shape[4]
    U6 = U43.reshape(Cin // 8, 2, nblk, 6, 3, npp, 4, 16, 2, 2)                        # c, half, blk, a, b3, np, kg, lt, par, s
    U6 = U6.permute(3, 1, 4, 2, 5, 8, 7, 0, 6, 9).reshape(6, 6, Cout, Cin).to(dt)      # [a][b = 3 half + b3][co = blk, np, par, lt][ci = c, kg, s]
    M = torch.einsum("ntuabc,aboc->ntuabo", V, U6)
    Y = torch.einsum("pa,ntuabo,qb->ntupqo", AT, M, AT)                               # [N,Ht,Wt,4,4,Cout]
    y = Y.permute(0, 1, 3, 2, 4, 5).reshape(N, 4 * Ht, 4 * Wt, Cout)[:, :H, :W] + (bias.to(dt) if bias is not None else 0)
    if relu:
        y = F.relu(y)
    yp = F.max_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous() if pool else None
    return (y.contiguous() if full else None), yp


def wino43_conv3x3_multi(xs, U43, bias, relu=True, full=True, pool=False):
    res = [_wino43(x, U43, bias, relu, full, pool) for x in xs]
    return ([r[0] for r in res] if full else None), ([r[1] for r in res] if pool else None)


def corr2d_wino43_multi(xs, U43, outs, kblocks=5, k_true=None):
    """As corr2d_wino_multi with the F(4x4,3x3) blocks of backbone.winograd43_corr_filters."""
    kb = kblocks
    for x, o in zip(xs, outs):
        N, _, H, W, Cin = x.shape
        nc = Cin // 8
        acc = torch.zeros((N, H, W, U43.shape[2] * U43.shape[4] * 32), dtype=x.dtype)
        pad = 3 * (kb - 1) // 2 + 1
        xp = F.pad(x[:, 0], (0, 0, pad, pad, pad, pad))
        for bi in range(kb):
            for bj in range(kb):
                b = bi * kb + bj
                sh = xp[:, 3 * bi:3 * bi + H + 2, 3 * bj:3 * bj + W + 2]
                y, _ = _wino43(sh.contiguous(), U43.view(nc, kb * kb, *U43.shape[1:])[:, b].contiguous(), None, False, True, False)
                acc += y[:, 1:H + 1, 1:W + 1]
        o.copy_(acc[:, None])
    return outs


def wino16_conv3x3_multi(xs, U16, bias, relu=True, full=True, pool=False):
    """The 16-bit kernel's arithmetic restated: F(2x2,3x3) with the host-rounded filters U16 and the transformed input ROUNDED to the
    same type before the products (fp32 transform, exact products of 16-bit values, wide accumulation)."""
    dt16 = U16.dtype
    Cout = U16.shape[2]
    U = U16.double().clone()
    swap = (torch.arange(Cout) & 8) != 0
    U[:, :, swap] = torch.cat([U[:, :, swap, 8:], U[:, :, swap, :8]], -1)                # undo the LDS swizzle
    outs = []
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    for x in xs:
        N, H, W, Cin = x.shape
        Ht, Wt = (H + 1) // 2, (W + 1) // 2
        xp = F.pad(x.float(), (0, 0, 1, 2 * Wt + 1 - W, 1, 2 * Ht + 1 - H))
        d = xp.unfold(1, 4, 2).unfold(2, 4, 2)                                           # [N,Ht,Wt,C,4,4]
        V = torch.einsum("ai,ntucij,bj->ntuabc", BT, d, BT).to(dt16).double()            # fp32 transform, rounded operand
        U4 = U.permute(1, 2, 0, 3).reshape(4, 4, Cout, Cin)                              # [a][b][co][chunk*16 + k]
        M = torch.einsum("ntuabc,aboc->ntuabo", V, U4)
        Y = torch.einsum("pa,ntuabo,qb->ntupqo", AT, M, AT)
        y = Y.permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * Ht, 2 * Wt, Cout)[:, :H, :W] + bias.double()
        if relu:
            y = F.relu(y)
        yp = F.max_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous() if pool else None
        outs.append((y.to(x.dtype).contiguous() if full else None, yp.to(x.dtype) if pool else None))
    return ([o[0] for o in outs] if full else None), ([o[1] for o in outs] if pool else None)


def l2norm_rows(x):
    x.copy_(F.normalize(x, dim=-1))
    return x


def nchw_to_nhwc(x, out, l2norm):
    v = F.normalize(x, dim=1) if l2norm else x
    out.copy_(v.permute(0, 2, 3, 1).unsqueeze(1))
    return out


def selector_ref_sums(refs):
    r = refs.double()
    return r.sum(0), (r * r).sum(0)


def selector_prod_affine(que, r1, r2, D, eps=1e-5):
    n = D * que.shape[0]
    q = que.double()
    mean = (q * r1).sum(0) / n
    var = ((q * q * r2).sum(0) / n - mean * mean).clamp_min(0)
    rs = 1.0 / torch.sqrt(var + eps)
    return rs.float()[None], (-mean * rs).float()[None]


def selector_scan(que, refs):
    smap = (refs * que[None]).sum(2)
    vps = (smap * (smap / smap.max(1, keepdim=True)[0])).sum(1)
    return smap, vps


def refiner_volume_kp(feats, ref_Ks, ref_poses, K_in, pose_in, lin, h_in, w_in, mean_in, std):
    if feats.dim() == 5:                                            # a batch of queries
        for b in range(feats.shape[0]):
            refiner_volume_kp(feats[b], ref_Ks[b], ref_poses[b], K_in[b], pose_in[b], lin, h_in, w_in, mean_in[b], std[b])
        return mean_in, std
    projs = torch.cat([ref_Ks @ ref_poses, (K_in @ pose_in)[None]], 0)
    return refiner_volume(feats, projs, pose_in[:, :3], lin, h_in, w_in, mean_in, std)


def selector_levels(ques, refs, sums, Dg, eps=1e-5, want_maps=False):
    if ques[0].dim() == 3:                                          # [qn,HW,C]: a batch of queries
        per = [selector_levels([q[i] for q in ques], refs, sums, Dg, eps, want_maps) for i in range(ques[0].shape[0])]
        maps = [torch.stack([p[3][l] for p in per], 0) for l in range(len(ques))] if want_maps else None
        return torch.stack([p[0] for p in per], 0), torch.stack([p[1] for p in per], 0), torch.stack([p[2] for p in per], 0), maps
    res = [selector_scan(q, r) for q, r in zip(ques, refs)]
    aff = [selector_prod_affine(q, s[0], s[1], Dg, eps) for q, s in zip(ques, sums)]
    return (torch.stack([r[1] for r in res], 0), torch.cat([a[0] for a in aff], 0), torch.cat([a[1] for a in aff], 0),
            [r[0] for r in res] if want_maps else None)


def refiner_volume(feats, projs, rot_in, lin, h_in, w_in, mean_in, std):
    V, fh, fw, C = feats.shape
    sn = lin.numel()
    g = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, sn ** 3, 3) @ rot_in
    X = g @ projs[:, :, :3].permute(0, 2, 1) + projs[:, :, 3:].permute(0, 2, 1)      # V,n,3
    z = X[..., 2:].clone(); z[z < 1e-4] = 1e-4
    uv = X[..., :2] / z
    gx = ((uv[..., 0] + 0.5) / w_in - 0.5) * 2
    gy = ((uv[..., 1] + 0.5) / h_in - 0.5) * 2
    grid = torch.stack([gx, gy], -1).reshape(V, 1, -1, 2)
    vol = F.grid_sample(feats.permute(0, 3, 1, 2), grid, mode="bilinear", padding_mode="zeros", align_corners=False)[:, :, 0]
    ref = vol[:-1]
    mean_in[:, :C] = ref.mean(0).T
    mean_in[:, C:] = vol[-1].T
    std.copy_(ref.std(0).T if V > 2 else torch.zeros_like(std))
    return mean_in, std


def resize_bilinear_pyramid(imgs, sizes):
    """F.interpolate(..., mode='bilinear') per size (reference network/detector.py:236-241)."""
    return [imgs if tuple(sz) == tuple(imgs.shape[2:]) else F.interpolate(imgs, size=tuple(sz), mode="bilinear") for sz in sizes]


def detector_assemble(s0, s1, s2, hc, wc, mu_sigma, clip, hs, ws, scale_idx, stacked, batch=1):
    if batch > 1:
        for b in range(batch):
            cut = lambda t, n: t[b * n:(b + 1) * n]
            detector_assemble(cut(s0, hc * wc), cut(s1, (hc // 2) * (wc // 2)), cut(s2, (hc // 4) * (wc // 4)), hc, wc, mu_sigma, clip,
                              hs, ws, scale_idx, cut(stacked, hs * ws))
        return stacked
    rfn = s0.shape[1]
    maps = []
    for l, s in enumerate((s0, s1, s2)):
        m = s.reshape(hc >> l, wc >> l, rfn).permute(2, 0, 1)[None]
        if l: m = F.interpolate(m, scale_factor=2 ** l)
        maps.append(torch.clip((m - mu_sigma[l][0]) / mu_sigma[l][1], -clip, clip))
    m = torch.cat(maps, 0)                                           # 3,rfn,hc,wc
    m = F.interpolate(m, size=(hs, ws), mode="bilinear", align_corners=False)
    stacked[:, :, 3 * scale_idx:3 * scale_idx + 3] = m.permute(2, 3, 1, 0).reshape(hs * ws, rfn, 3)
    return stacked


def detector_score_mlp_max(stacked, w0, b0, w1, b1):
    h = F.relu(stacked @ w0.T + b0)
    return (h @ w1.T + b1).max(1)[0]


def detector_decode(scores, offset, scale, hs, ws, pool_ratio, batch=1):
    if batch > 1:
        P = hs * ws
        return torch.stack([detector_decode(scores[b * P:(b + 1) * P], offset[b * P:(b + 1) * P], scale[b * P:(b + 1) * P], hs, ws, pool_ratio)
                            for b in range(batch)], 0)
    idx = int(torch.argmax(scores[:, 0]))
    x, y = idx % ws, idx // ws
    res = torch.empty(5, dtype=torch.float32, device=scores.device)
    res[0] = (x + offset[idx, 0] + 0.5) * pool_ratio - 0.5
    res[1] = (y + offset[idx, 1] + 0.5) * pool_ratio - 0.5
    res[2] = 2 ** scale[idx, 0]
    res[3], res[4] = x, y
    return res


def vps_norm(vps, feats, c_off):
    if vps.dim() == 3:                                              # [qn,3,D]
        D = vps.shape[-1]
        for b in range(vps.shape[0]):
            vps_norm(vps[b], feats[b * D:(b + 1) * D], c_off)
        return feats
    feats[:, c_off:c_off + 3] = F.instance_norm(vps[None], eps=1e-5)[0].T
    return feats


def max_an_add(x, rfn, an, embed, out, batch=1):
    out.copy_((x.reshape(batch, rfn, an, -1).max(2)[0] + embed[None]).reshape(batch * rfn, -1))
    return out


def attention(q, k, v, heads, out, batch=1):
    if batch > 1:
        n = q.shape[0] // batch
        for b in range(batch):
            sl = slice(b * n, (b + 1) * n)
            attention(q[sl], k[sl], v[sl], heads, out[sl])
        return out
    n, C = q.shape
    dh = C // heads
    qh, kh, vh = (t.reshape(n, dh, heads) for t in (q, k, v))
    s = torch.einsum("ndh,mdh->hnm", qh, kh) / dh ** 0.5
    o = torch.einsum("hnm,mdh->ndh", F.softmax(s, -1), vh)
    out.copy_(o.reshape(n, C))
    return out


def layernorm(x, gamma, beta, out, eps=1e-5):
    out.copy_(F.layer_norm(x, (x.shape[1],), gamma, beta, eps))
    return out


def affine_act_add(x, out, scale=None, shift=None, relu=False, residual=None, rows_per_group=0):
    v = x
    if scale is not None:
        if rows_per_group:
            g = torch.arange(x.shape[0]) // rows_per_group
            v = v * scale.reshape(-1, x.shape[1])[g] + shift.reshape(-1, x.shape[1])[g]
        else:
            v = v * scale.reshape(1, -1) + shift.reshape(1, -1)
    if relu: v = F.relu(v)
    if residual is not None: v = v + residual
    out.copy_(v)
    return out


def linear_gemv(x, W, bias, act=0):
    return _act(F.linear(x, W, bias), act)


def warp_perspective(src_u8, H, dh, dw, out_float=False):
    """Plain bilinear inverse warp with zero border (float weights)."""
    import numpy as np
    H = np.asarray(H, dtype=np.float64)
    if H.shape == (2, 3): H = np.concatenate([H, [[0.0, 0.0, 1.0]]], 0)
    hinv = torch.from_numpy(np.linalg.inv(H).astype(np.float32))
    sh, sw, ch = src_u8.shape
    ys, xs = torch.meshgrid(torch.arange(dh, dtype=torch.float32), torch.arange(dw, dtype=torch.float32), indexing="ij")
    p = torch.stack([xs, ys, torch.ones_like(xs)], -1) @ hinv.T
    fx, fy = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
    gx, gy = (fx + 0.5) / sw * 2 - 1, (fy + 0.5) / sh * 2 - 1
    img = src_u8.float().permute(2, 0, 1)[None].cpu()
    out = F.grid_sample(img, torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="zeros", align_corners=False)[0]
    out = out.permute(1, 2, 0)
    out = (out / 255.0) if out_float else out.round().clamp(0, 255).to(torch.uint8)
    return out.to(src_u8.device)


# ---- device-resident chain ops emulated with the HOST pose algebra (gen6d_amd/geometry.py)
def _np(t):
    import numpy as np
    return t.detach().cpu().numpy().astype(np.float64)


def chain_crop_from_detection(det, size):
    import numpy as np
    from gen6d_amd import geometry as G
    if det.dim() == 2:
        return torch.cat([chain_crop_from_detection(d_, size) for d_ in det], 0)
    d = _np(det)
    M = np.concatenate([G.crop_transform(d[:2], 1 / d[2], 0, size), [[0, 0, 1]]], 0)
    return torch.from_numpy(np.linalg.inv(M).reshape(1, 9).astype(np.float32))


def chain_pose_from_selection(det, logits, angles, ref_poses, ref_Ks, que_K, center):
    import numpy as np
    from gen6d_amd import geometry as G
    if det.dim() == 2:
        rs = [chain_pose_from_selection(det[b], logits[b], angles[b], ref_poses, ref_Ks, que_K[b], center) for b in range(det.shape[0])]
        return torch.stack([r[0] for r in rs], 0), torch.stack([r[1] for r in rs], 0)
    d, lg, an = _np(det), _np(logits), _np(angles)
    i = int(np.argmax(lg))
    pose = G.estimate_pose_from_similarity_transform_compose(d[:2], d[2], an[i], _np(ref_poses)[i].reshape(3, 4), _np(ref_Ks)[i].reshape(3, 3),
                                                             _np(que_K).reshape(3, 3), _np(center))
    return torch.from_numpy(pose.astype(np.float32)), torch.tensor([float(i), float(an[i])])


def chain_refine_prepare(pose_in, que_K, norm, size, margin, sub_poses, sub_Ks, ref_num, angle_step=0.0):
    import numpy as np
    from gen6d_amd import geometry as G
    if pose_in.dim() == 2:
        rs = [chain_refine_prepare(pose_in[b], que_K[b], norm, size, margin, sub_poses, sub_Ks, ref_num, angle_step) for b in range(pose_in.shape[0])]
        return tuple(torch.stack([r[k] for r in rs], 0) for k in range(len(rs[0])))
    nm = _np(norm); c0 = np.zeros(3)
    in_pose = G.normalize_pose(_np(pose_in).reshape(3, 4), nm[0], nm[1:]).astype(np.float64)
    K = _np(que_K).reshape(3, 3)
    _, new_f = G.let_me_look_at(in_pose, K, c0)
    scale = size * (1 - margin) / 2.0 * np.linalg.norm(G.pose_inverse(in_pose)[:, 3]) / new_f
    K_warp, pose_warp, pose_rect, H = G.look_at_crop_params(K, in_pose, G.project_points(c0[None], in_pose, K)[0][0], 0, scale, size, size)
    sp, sk = _np(sub_poses).reshape(-1, 3, 4), _np(sub_Ks).reshape(-1, 3, 3)
    idx = np.argsort(-G.view_correlation(pose_warp[None].astype(np.float64), sp, c0)[0], kind="stable")[:ref_num]
    Ks, poses, hinvs, buckets = [], [], [np.linalg.inv(H)], []
    for i in idx:
        cen = G.project_points(c0[None], sp[i], sk[i])[0][0]
        f_look = G.let_me_look_at(sp[i], sk[i], c0)[1]
        _, ang = G.scale_rotation_difference_from_cameras(sp[i][None], pose_warp[None].astype(np.float64), sk[i][None],
                                                          K_warp[None].astype(np.float64), c0)
        a = float(ang[0])
        if angle_step > 0:
            buckets.append(int(np.floor(a / angle_step + 0.5))); a = buckets[-1] * angle_step
        Kn, pn, _, Hr = G.look_at_crop_params(sk[i], sp[i], cen, a, size * (1 - margin) / 2.0 * np.linalg.norm(G.pose_inverse(sp[i])[:, 3]) / f_look,
                                              size, size)
        Ks.append(Kn); poses.append(pn); hinvs.append(np.linalg.inv(Hr))
    geo = np.concatenate([np.ravel(K_warp), np.ravel(pose_warp), np.ravel(pose_rect), np.ravel(Ks), np.ravel(poses), np.ravel(hinvs)])
    if angle_step > 0:
        return torch.from_numpy(geo.astype(np.float32)), torch.from_numpy(idx.astype(np.int32)), torch.tensor(buckets, dtype=torch.int32)
    return torch.from_numpy(geo.astype(np.float32)), torch.from_numpy(idx.astype(np.int32))


def chain_refine_update(rot, off, scl, geo, norm):
    import numpy as np
    from gen6d_amd import geometry as G
    if geo.dim() == 2:
        return torch.stack([chain_refine_update(rot[b], off[b], scl[b], geo[b], norm) for b in range(geo.shape[0])], 0)
    g, nm, c0 = _np(geo), _np(norm), np.zeros(3)
    K_warp, pose_warp, pose_rect = g[:9].reshape(3, 3), g[9:21].reshape(3, 4), g[21:33].reshape(3, 4)
    sim = G.compose_sim_pose(2 ** float(_np(scl)[0]), _np(rot), _np(off), pose_warp, c0)
    pr = G.pose_compose(G.pose_sim_to_pose_rigid(sim, pose_warp, K_warp, K_warp, c0), G.pose_inverse(pose_rect))
    return torch.from_numpy(G.denormalize_pose(pr, nm[0], nm[1:]).astype(np.float32))


def warp_batch(stack, single, idx, hinv, dh, dw, out=None):
    import numpy as np
    B = hinv.shape[0]
    res = []
    for b in range(B):
        sel = -1 if idx is None else int(idx[b])
        src = single if sel < 0 else stack[sel]
        H = np.linalg.inv(_np(hinv[b]).reshape(3, 3))
        res.append(warp_perspective(src, H, dh, dw).float().div(255).permute(2, 0, 1))
    r = torch.stack(res, 0)
    if out is not None:
        out.copy_(r)
        return out
    return r


# ---- the direct kernel on 16-bit activations (ops.conv16_*): references on the operands' own values -------------------------------
_T16 = {1: torch.bfloat16, 2: torch.float16, 3: torch.float16}


def _to_pairs(v):
    hi = v.to(torch.float16)
    lo = (v - hi.to(v.dtype)).to(torch.float16)
    return torch.stack([hi, lo], -2)


def vgg_conv1_pool_nhwc16(x, w_oihw, bias, out=None, norm=None, mode=None):
    from gen6d_amd import ops
    mode = ops.MATH_MODE if mode is None else mode
    y = vgg_conv1_pool_nhwc(x, w_oihw, bias, norm=norm)
    y = _to_pairs(y) if mode == 3 else y.to(_T16[mode])
    if out is not None:
        out.copy_(y)
        return out
    return y


class _RefConv16Filters:
    def __init__(self, w_taps, mode, layout):
        self.w, self.mode, self.layout = w_taps, mode, layout
        self.Cout, self.taps, self.Cin = w_taps.shape
        self.acc_scale = 1.0
        self.data = w_taps


def conv16_pack(w_taps, mode, layout=1):
    """Reference: keeps the fp32 taps (mode 1 / 2: rounded to the 16-bit type, as the kernel's operands are)."""
    w = w_taps if mode == 3 else w_taps.to(_T16[mode]).to(w_taps.dtype)
    return _RefConv16Filters(w, mode, layout)


def conv16_direct_multi(xs, filt, bias, relu=True, full=None, pool=None, kd=1, stats=None, rows_per_group=0, out_full=None):
    """Reference of g6d_conv16_direct_multi on the VALUES of the operands (pairs: hi + lo), F.conv2d / F.conv3d in the filters' dtype."""
    pair = filt.mode == 3
    dt = filt.w.dtype if filt.w.dtype in (torch.float32, torch.float64) else torch.float32
    fulls, pools = [], []
    for x in xs:
        v = (x[..., 0, :].to(dt) + x[..., 1, :].to(dt)) if pair else x.to(dt)
        if kd == 1:
            w = filt.w.to(dt).reshape(filt.Cout, 3, 3, filt.Cin).permute(0, 3, 1, 2)
            y = F.conv2d(v.permute(0, 3, 1, 2), w, None if bias is None else bias.to(dt), padding=1).permute(0, 2, 3, 1)
        else:
            w = filt.w.to(dt).reshape(filt.Cout, 3, 3, 3, filt.Cin).permute(0, 4, 1, 2, 3)
            y = F.conv3d(v.permute(0, 4, 1, 2, 3), w, None if bias is None else bias.to(dt), padding=1).permute(0, 2, 3, 4, 1)
        if stats is not None:
            g = y.reshape(-1, y.shape[-1]) if rows_per_group <= 0 else y.reshape(-1, rows_per_group, y.shape[-1])
            g = g[None] if rows_per_group <= 0 else g
            stats[:g.shape[0], :, 0] += g.double().sum(1)
            stats[:g.shape[0], :, 1] += (g.double() ** 2).sum(1)
        if relu:
            y = F.relu(y)

        def coded(t, kind):
            if kind is None:
                return None
            if kind == "t16":
                return _to_pairs(t) if pair else t.to(_T16[filt.mode])
            return t.to(torch.float32).contiguous()
        if out_full is not None:
            out_full[len(fulls)].copy_(coded(y, full).reshape(out_full[len(fulls)].shape))
            fulls.append(out_full[len(fulls)])
        else:
            fulls.append(coded(y, full))
        pools.append(coded(F.max_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1), pool) if pool is not None else None)
    return fulls, pools


def corr16_pack(w_taps, mode):
    f = conv16_pack(w_taps, mode, 1)
    f.k = int(round(w_taps.shape[1] ** 0.5))
    return f


def corr16_multi(xs, filt, outs):
    pair = filt.mode == 3
    dt = filt.w.dtype if filt.w.dtype in (torch.float32, torch.float64) else torch.float32
    k = filt.k
    for x, o in zip(xs, outs):
        v = (x[..., 0, :].to(dt) + x[..., 1, :].to(dt)) if pair else x.to(dt)
        w = filt.w.to(dt).reshape(filt.Cout, k, k, filt.Cin).permute(0, 3, 1, 2)
        y = F.conv2d(v.permute(0, 3, 1, 2), w, None, padding=k // 2).permute(0, 2, 3, 1)
        o.copy_(y.reshape(o.shape).to(o.dtype))
    return outs


def product_split16(ref, que, scale, shift, mode):
    v = (ref[None] * que[:, None]) * scale[:, None, None, :] + shift[:, None, None, :]            # [qn, D, P, C]
    v = v.reshape(-1, v.shape[2], v.shape[3])
    return _to_pairs(v) if mode == 3 else v.to(_T16[mode])


def affine_split16(x, scale, shift, per_n, relu, pool, mode):
    N, D, H, W, C = x.shape
    out = torch.empty((N, 1, H // 2, W // 2, C) if pool else (N, 1, H, W, C), dtype=x.dtype)
    affine_act_pool(x, out, scale, shift, per_n=per_n, relu=relu, pool=1 if pool else 0)
    v = out[:, 0]
    return _to_pairs(v) if mode == 3 else v.to(_T16[mode])


def patch_ops(monkeypatch):
    """Route gen6d_amd.ops.* to the references above (CPU host-logic tests only)."""
    import sys
    from gen6d_amd import ops
    me = sys.modules[__name__]
    for name in dir(ops):
        if not name.startswith("_") and callable(getattr(ops, name)) and hasattr(me, name):
            monkeypatch.setattr(ops, name, getattr(me, name))
