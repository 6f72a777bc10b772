"""CPU ORACLE — test infrastructure, not product code.

A stateless, functional restatement (torch CPU ops on a plain ``state_dict``) of the tensor hot path of
liuyuan-pal/Gen6D: detector score-map correlation, selector viewpoint similarity + in-plane rotation, refiner
feature-volume pose update.  Each function cites the reference lines it follows.  It exists to CHECK the HIP path;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product package
(gen6d_amd/) never does.

Parity pinning: the reference has no tests or golden vectors; the oracle is pinned against outputs
of the reference's own modules, run in the build container with import stubs by tests/golden/make_golden.py and
committed as tests/golden/*.npz (tests/test_oracle_golden.py).  It accepts float64 state_dicts/inputs to provide
the fp64 ground truth used to grade fp32 noise.
"""

import time as _time

import numpy as np
import torch
import torch.nn.functional as F

TIMERS = None      # set to a dict by bench.py's cpu_baseline leg to split out the torch.std share (seconds, accumulated)

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
VGG_CONVS = (0, 4, 8, 11, 15, 18, 22, 25)          # conv indices in features; BN at +1
VGG_POOL_BEFORE = {4, 8, 15, 22}                   # a 2x2 max-pool precedes these convs
SCORE_STATS = ((36.264317, 13.151907), (13910.291, 5345.965), (829.70807, 387.98788))  # detector.py:138


def img_norm(x):
    """torchvision Normalize as used at detector.py:156,189 / selector.py:22,115 / refiner.py:62,65."""
    m = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - m) / s


def vgg_stages(sd, prefix, x):
    """VGG-11-BN (eval BN). Returns feature maps after each of the 8 conv blocks, *pre*- and *post*-ReLU.

    pretrain_models.py:86-111: conv3x3 -> BN -> ReLU, pools before convs 4, 8, 15, 22 (and a last pool at 28).
    """
    pre, post = {}, {}
    for i in VGG_CONVS:
        if i in VGG_POOL_BEFORE:
            x = F.max_pool2d(x, 2, 2)
        x = F.conv2d(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"], padding=1)
        x = F.batch_norm(x, sd[f"{prefix}.{i + 1}.running_mean"], sd[f"{prefix}.{i + 1}.running_var"],
                         sd[f"{prefix}.{i + 1}.weight"], sd[f"{prefix}.{i + 1}.bias"], False, 0.0, 1e-5)
        pre[i] = x
        x = F.relu(x)
        post[i] = x
    return pre, post


def vgg_x012(sd, prefix, x):
    """VGGBNPretrain.forward (pretrain_models.py:17-25): x0 @1/8 post-ReLU, x1 @1/16 = BN output of conv 25
    WITHOUT its ReLU (split (21,27) ends at the BatchNorm), x2 = max-pool of x1."""
    pre, post = vgg_stages(sd, prefix, x)
    x0, x1 = post[18], pre[25]
    return x0, x1, F.max_pool2d(x1, 2, 2)


def vgg_v3(sd, prefix, x):
    """VGGBNPretrainV3.forward (pretrain_models.py:66-72): x2 @1/4 (256), x3 @1/8 (512), x4 @1/16 pre-ReLU."""
    pre, post = vgg_stages(sd, prefix, x)
    return post[11], post[18], pre[25]


# ------------------------------------------------------------------------------------------------ detector
def detector_ref_feats(sd, ref_imgs):
    """Detector.load_impl (detector.py:199-205): nearest resize to 120x120, ImageNet norm, VGG."""
    ref_imgs = F.interpolate(ref_imgs, size=(120, 120))
    return vgg_x012(sd, "backbone.features", img_norm(ref_imgs))


def detector_scores(sd, que_imgs, ref_feats):
    """Detector.get_scores + normalize_scores (detector.py:207-230)."""
    q0, q1, q2 = vgg_x012(sd, "backbone.features", img_norm(que_imgs))
    r0, r1, r2 = ref_feats
    s2 = F.conv2d(q2, r2, padding=1)
    s1 = F.conv2d(q1, r1, padding=3)
    s0 = F.conv2d(q0, r0, padding=7)
    s2 = F.interpolate(s2, scale_factor=4)
    s1 = F.interpolate(s1, scale_factor=2)
    out = []
    for s, (mu, sigma) in zip((s0, s1, s2), SCORE_STATS):
        out.append(torch.clip((s - mu) / sigma, min=-10, max=10))
    return torch.stack(out, 1)                                     # qn,3,rfn,h/8,w/8


def _head(sd, name, x):
    x = F.relu(F.conv2d(x, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], padding=1))
    x = F.relu(F.conv2d(x, sd[f"{name}.2.weight"], sd[f"{name}.2.bias"], padding=1))
    return F.conv2d(x, sd[f"{name}.4.weight"], sd[f"{name}.4.bias"], padding=1)


def detector_detect(sd, que_imgs, ref_feats, scales=(-1.0, -0.5, 0.0, 0.5), return_intermediates=False):
    """Detector.detect_impl (detector.py:232-266)."""
    qn, _, hq, wq = que_imgs.shape
    hs, ws = hq // 8, wq // 8
    maps = []
    for s in scales:
        ht, wt = int(np.round(hq * 2 ** s)), int(np.round(wq * 2 ** s))
        if ht % 32 != 0: ht = (ht // 32 + 1) * 32
        if wt % 32 != 0: wt = (wt // 32 + 1) * 32
        cur = F.interpolate(que_imgs, size=(ht, wt), mode="bilinear")
        sc = detector_scores(sd, cur, ref_feats)
        _, _, rfn, hc, wc = sc.shape
        maps.append(F.interpolate(sc.reshape(qn, 3 * rfn, hc, wc), size=(hs, ws), mode="bilinear")
                    .reshape(qn, 3, rfn, hs, ws))
    stacked = torch.cat(maps, 1)                                   # qn,12,rfn,hs,ws
    x = F.conv3d(stacked, sd["score_conv.0.weight"], sd["score_conv.0.bias"])
    x = F.conv3d(F.relu(x), sd["score_conv.2.weight"], sd["score_conv.2.bias"])
    feats = torch.max(x, 2)[0]                                     # qn,64,hs,ws
    scores = _head(sd, "score_predict", feats)
    offset = _head(sd, "offset_predict", feats)
    scale = _head(sd, "scale_predict", feats)
    out = {"scores": scores, "select_pr_offset": offset, "select_pr_scale": scale, "pool_ratio": 8}
    flat = torch.argmax(scores.flatten(1), 1)                      # detector.py:84-95 (one channel)
    sy, sx = flat // ws, flat % ws
    out["que_select_id"] = torch.stack([sx, sy], 1)
    if return_intermediates:
        out["stacked"], out["score_feats"] = stacked, feats
    return out


def detector_parse(out):
    """BaseDetector.parse_detection (detector.py:97-121): positions [qn,2] (x,y px), scales [qn]."""
    qn = out["scores"].shape[0]
    sx, sy = out["que_select_id"][:, 0], out["que_select_id"][:, 1]
    ar = torch.arange(qn)
    pos = torch.stack([sx, sy], -1) + out["select_pr_offset"][ar, :, sy, sx]
    pos = (pos + 0.5) * out["pool_ratio"] - 0.5
    return pos, 2 ** out["select_pr_scale"][ar, 0, sy, sx]


# ------------------------------------------------------------------------------------------------ selector
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def selector_ref_state(sd, ref_imgs, ref_poses, object_center, object_vert):
    """ViewpointSelector.extract_ref_feats, eval branch (selector.py:121-148).

    ref_imgs [an,rfn,3,h,w] -> cache of 3 tensors [an,rfn,512,h_l,w_l] (L2-normalised over C), pose embed [rfn,512].
    """
    an, rfn, _, h, w = ref_imgs.shape
    feats = vgg_x012(sd, "backbone.features", img_norm(ref_imgs.reshape(an * rfn, 3, h, w)))
    cache = [F.normalize(f, dim=1).reshape(an, rfn, *f.shape[1:]) for f in feats]
    cam = (-ref_poses[:, :3, :3].permute(0, 2, 1) @ ref_poses[:, :3, 3:])[..., 0] - object_center[None]
    fwd = cam[0]
    y = torch.linalg.cross(object_vert, fwd)
    x = torch.linalg.cross(y, object_vert)
    R = torch.stack([F.normalize(x, dim=0), F.normalize(y, dim=0), F.normalize(object_vert, dim=0)], 0)
    v = F.normalize(cam @ R.T, dim=1)
    e = F.relu(F.linear(v, sd["view_point_encoder.0.weight"], sd["view_point_encoder.0.bias"]))
    e = F.relu(F.linear(e, sd["view_point_encoder.2.weight"], sd["view_point_encoder.2.bias"]))
    e = F.linear(e, sd["view_point_encoder.4.weight"], sd["view_point_encoder.4.bias"])
    return cache, e


# corr_conv_list[l]: (conv index, followed-by-IN, followed-by-ReLU, followed-by-MaxPool)  selector.py:27-69
_CORR_LAYERS = (
    ((1, 1, 1, 0), (4, 1, 0, 1), (7, 1, 1, 0), (10, 1, 0, 1), (13, 1, 1, 0), (16, 0, 0, 0)),
    ((1, 1, 1, 0), (4, 1, 0, 1), (7, 1, 1, 0), (10, 0, 0, 0)),
    ((1, 1, 1, 0), (4, 0, 0, 0)),
)


def _attention_block(sd, p, x):
    """AttentionBlock(512,512,512,8,skip_connect=False).forward(x,x) (attention.py:4-17,50-68)."""
    b, f, n = x.shape
    q = F.conv1d(x, sd[f"{p}.conv_query.weight"], sd[f"{p}.conv_query.bias"]).reshape(b, 64, 8, n)
    k = F.conv1d(x, sd[f"{p}.conv_key.weight"], sd[f"{p}.conv_key.bias"]).reshape(b, 64, 8, n)
    v = F.conv1d(x, sd[f"{p}.conv_feats.weight"], sd[f"{p}.conv_feats.bias"]).reshape(b, 64, 8, n)
    s = torch.einsum("bdhn,bdhm->bhnm", q, k) / 64 ** 0.5
    o = torch.einsum("bhnm,bdhm->bdhn", F.softmax(s, dim=-1), v).reshape(b, 512, n)
    o = F.conv1d(o, sd[f"{p}.conv_merge.weight"], sd[f"{p}.conv_merge.bias"])
    o = F.layer_norm(o.permute(0, 2, 1), (512,), sd[f"{p}.norm.norm.weight"], sd[f"{p}.norm.norm.bias"])
    return o.permute(0, 2, 1)


def selector_forward(sd, que_imgs, cache, pose_embed, return_intermediates=False):
    """ViewpointSelector.compute_view_point_feats (selector.py:177-215) -> logits [qn,rfn], angles [qn,rfn]."""
    que = [F.normalize(f, dim=1) for f in vgg_x012(sd, "backbone.features", img_norm(que_imgs))]
    vps, corr = [], []
    inter = {}
    for l, (ref, q) in enumerate(zip(cache, que)):
        ref = ref.permute(1, 0, 2, 3, 4)                               # rfn,an,f,h,w
        prod = q[:, None, None] * ref[None]                            # qn,rfn,an,f,h,w
        qn, rfn, an, f, h, w = prod.shape
        prod = prod.permute(0, 3, 1, 2, 4, 5).reshape(qn, f, rfn * an, h, w)
        x = _inorm(prod)
        for idx, has_in, has_relu, has_pool in _CORR_LAYERS[l]:
            p = f"corr_conv_list.{l}.{idx}"
            x = F.conv3d(x, sd[p + ".weight"], sd[p + ".bias"], padding=(0, 1, 1))
            if has_in: x = _inorm(x)
            if has_relu: x = F.relu(x)
            if has_pool: x = F.max_pool3d(x, (1, 2, 2), (1, 2, 2))
        corr.append(x)                                                 # qn,256,D,4,4
        smap = prod.sum(1).flatten(2)                                  # qn,D,hw
        vps.append((smap * (smap / smap.max(2, keepdim=True)[0])).sum(2).reshape(qn, rfn, an))
        if return_intermediates:
            inter[f"score_map{l}"] = smap
    x = torch.cat(corr, 1)
    x = F.conv3d(x, sd["corr_feats_conv.0.weight"], sd["corr_feats_conv.0.bias"])
    x = F.relu(_inorm(x))
    x = F.conv3d(x, sd["corr_feats_conv.3.weight"], sd["corr_feats_conv.3.bias"])
    x = F.avg_pool3d(x, (1, 4, 4))[..., 0, 0].reshape(qn, 512, rfn, an)
    vpsn = _inorm(torch.stack(vps, 1))                                 # InstanceNorm2d(3) over (rfn,an)
    feats = torch.cat([x, vpsn], 1)                                    # qn,515,rfn,an
    s = F.conv2d(feats, sd["score_process.0.weight"], sd["score_process.0.bias"])
    s = F.conv2d(F.relu(s), sd["score_process.2.weight"], sd["score_process.2.bias"])
    s = s.max(3)[0] + pose_embed.T.unsqueeze(0)                        # qn,512,rfn
    for i in range(2):
        msg = _attention_block(sd, f"atts.{i}", s)
        y = F.conv1d(torch.cat([s, msg], 1), sd[f"mlps.{i}.0.weight"], sd[f"mlps.{i}.0.bias"])
        y = F.relu(_inorm(y))
        y = F.conv1d(y, sd[f"mlps.{i}.3.weight"], sd[f"mlps.{i}.3.bias"])
        s = F.relu(_inorm(y)) + s
    lg = F.conv1d(F.relu(F.conv1d(s, sd["score_predict.0.weight"], sd["score_predict.0.bias"])),
                  sd["score_predict.2.weight"], sd["score_predict.2.bias"])[:, 0]
    a = feats.permute(0, 1, 3, 2).reshape(qn, 515 * an, rfn)
    a = F.relu(F.conv1d(a, sd["angle_predict.0.weight"], sd["angle_predict.0.bias"]))
    a = F.relu(F.conv1d(a, sd["angle_predict.2.weight"], sd["angle_predict.2.bias"]))
    a = F.conv1d(a, sd["angle_predict.4.weight"], sd["angle_predict.4.bias"])[:, 0]
    if return_intermediates:
        inter.update(vps=torch.stack(vps, 1), corr_feats=x, feats=feats, score_feats=s)
        return lg, a, inter
    return lg, a


def selector_select(logits, angles):
    """ViewpointSelector.select_que_imgs tail (selector.py:172-175): raw angle at the arg-max reference."""
    idx = torch.argmax(logits, 1)
    return idx, angles[torch.arange(idx.shape[0]), idx]


# ------------------------------------------------------------------------------------------------ refiner


def refiner_feature_net(sd, imgs):
    """RefineFeatureNet.forward (refiner.py:64-78, layers :24-51) -> [n,128,h/4,w/4]."""
    x0, x1, x2 = [F.normalize(f, dim=1) for f in vgg_v3(sd, "feature_net.backbone.features", img_norm(imgs))]

    def pair(name, x):
        x = F.relu(_inorm(F.conv2d(x, sd[f"feature_net.{name}.0.weight"], sd[f"feature_net.{name}.0.bias"], padding=1)))
        return _inorm(F.conv2d(x, sd[f"feature_net.{name}.3.weight"], sd[f"feature_net.{name}.3.bias"], padding=1))

    y0 = pair("conv0", x0)
    y1 = F.interpolate(pair("conv1", x1), scale_factor=2, mode="bilinear")
    y2 = F.interpolate(pair("conv2", x2), scale_factor=4, mode="bilinear")
    return pair("conv_out", torch.cat([y0, y1, y2], 1))


def interpolate_volume_feats(feats, verts, projs, h_in, w_in):
    """VolumeRefiner.interpolate_volume_feats (refiner.py:183-206) with normalize_coords (operator.py:4-17).

    feats [b,f,h,w]; verts [b,n,3]; projs [b,3,4] -> [b,f,n]."""
    b, f, h, w = feats.shape
    X = verts @ projs[:, :, :3].permute(0, 2, 1) + projs[:, :, 3:].permute(0, 2, 1)
    z = X[..., 2:].clone()
    z[z < 1e-4] = 1e-4
    uv = X[..., :2] / z
    gx = ((uv[..., 0] + 0.5) / w_in - 0.5) * 2
    gy = ((uv[..., 1] + 0.5) / h_in - 0.5) * 2
    grid = torch.stack([gx, gy], -1).reshape(b, 1, -1, 2)
    return F.grid_sample(feats, grid, mode="bilinear", padding_mode="zeros", align_corners=False)[:, :, 0]


def refiner_volume(ref_feats, que_feats, poses_in, Ks_in, ref_poses, ref_Ks, h_in, w_in, sn=32):
    """VolumeRefiner.construct_feature_volume for qn=1 (refiner.py:208-247).

    ref_feats [rfn,f,h,w], que_feats [1,f,h,w] -> mean, std, in: [1,f,sn,sn,sn]."""
    dt = ref_feats.dtype
    g = torch.linspace(-1, 1, sn, dtype=torch.float32).to(dt)
    V = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, sn ** 3, 3)
    V = V @ poses_in[:, :3, :3]                                        # row vector times R_in
    rfn, f = ref_feats.shape[:2]
    proj = ref_Ks[0] @ ref_poses[0]                                    # rfn,3,4
    vol = interpolate_volume_feats(ref_feats, V.repeat(rfn, 1, 1), proj, h_in, w_in)
    mean = vol.mean(0)
    t0 = _time.perf_counter()
    std = vol.std(0)                                                   # unbiased over the refs
    if TIMERS is not None:      # torch.std over the ref axis is a pathological share of the CPU step
        TIMERS["refiner_std"] = TIMERS.get("refiner_std", 0.0) + _time.perf_counter() - t0
    qproj = Ks_in @ poses_in
    vin = interpolate_volume_feats(que_feats, V, qproj, h_in, w_in)
    shape = (1, f, sn, sn, sn)
    return mean.reshape(shape), std.reshape(shape), vin.reshape(shape)


def refiner_volume_net(sd, mean_in, std):
    """RefineVolumeEncodingNet.forward (refiner.py:136-143, layers :88-134)."""
    def c(p, x, stride=1):
        return F.conv3d(x, sd[f"volume_net.{p}.weight"], sd[f"volume_net.{p}.bias"], stride=stride, padding=1)

    a = c("mean_embed.3", F.relu(_inorm(c("mean_embed.0", mean_in))))
    b = c("var_embed.3", F.relu(_inorm(c("var_embed.0", std))))
    x = torch.cat([a, b], 1)
    x = F.relu(_inorm(c("conv0.0", x)))
    x = F.relu(_inorm(c("conv1.0", x, 2)))
    x = F.relu(_inorm(c("conv2.0", x)))
    x = F.relu(_inorm(c("conv3.0", x, 2)))
    x = F.relu(_inorm(c("conv4.0", x)))
    x = F.relu(_inorm(c("conv5.0", x, 2)))
    return c("conv5.3", x)                                             # 1,512,4,4,4


def refiner_regressor(sd, x):
    """RefineRegressor.forward (refiner.py:153-166)."""
    x = F.leaky_relu(F.linear(x, sd["regressor.fc.0.0.weight"], sd["regressor.fc.0.0.bias"]), 0.1)
    x = F.leaky_relu(F.linear(x, sd["regressor.fc.1.0.weight"], sd["regressor.fc.1.0.bias"]), 0.1)
    r = F.normalize(F.linear(x, sd["regressor.fcr.weight"], sd["regressor.fcr.bias"]), dim=1)
    t = F.linear(x, sd["regressor.fct.weight"], sd["regressor.fct.bias"])
    s = F.linear(x, sd["regressor.fcs.weight"], sd["regressor.fcs.bias"])
    return r, t, s


def refiner_forward(sd, que_imgs, Ks_in, poses_in, ref_imgs, ref_Ks, ref_poses, sn=32, return_intermediates=False):
    """VolumeRefiner.forward, inference branch, qn=1 (refiner.py:249-269).

    que_imgs [1,3,h,w]; ref_imgs [1,rfn,3,h,w]; Ks/poses as in the reference's data dict."""
    h_in, w_in = ref_imgs.shape[-2:]
    rf = refiner_feature_net(sd, ref_imgs[0])
    qf = refiner_feature_net(sd, que_imgs)
    mean, std, vin = refiner_volume(rf, qf, poses_in, Ks_in, ref_poses, ref_Ks, h_in, w_in, sn)
    x = refiner_volume_net(sd, torch.cat([mean, vin], 1), std)
    r, t, s = refiner_regressor(sd, x.flatten(1))
    out = {"rotation": r, "offset": t, "scale": s}
    if return_intermediates:
        out.update(ref_feats=rf, que_feats=qf, vol_mean=mean, vol_std=std, vol_in=vin, vol_code=x)
    return out


def to_double(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
