"""Parameter tables of the three Gen6D networks.

The drop-in must load the reference checkpoints unchanged, so every tensor keeps the key and shape the
reference's modules give it (reference: network/detector.py:147-184, network/selector.py:19-111,
network/refiner.py:24-62,88-134,153-160, network/pretrain_models.py:86-111).
Instead of re-declaring nn.Module trees, the whole contract is written once as (key-prefix, kind, shape) rows;
`gen6d_amd.network.params.ParamBank` turns a table into an object with an identical ``state_dict()``.
"""

# VGG-11-BN "features" Sequential: conv index -> (cin, cout); each conv at i is followed by a BatchNorm at i+1.
# Indices follow torchvision's layer numbering for cfg 'A' with batch-norm (reference pretrain_models.py:86-104).
VGG11_BN_CONVS = {0: (3, 64), 4: (64, 128), 8: (128, 256), 11: (256, 256),
                  15: (256, 512), 18: (512, 512), 22: (512, 512), 25: (512, 512)}
VGG11_BN_POOLS = (3, 7, 14, 21, 28)


def _vgg(prefix):
    rows = []
    for i, (ci, co) in VGG11_BN_CONVS.items():
        rows.append((f"{prefix}.{i}", "conv", (co, ci, 3, 3)))
        rows.append((f"{prefix}.{i + 1}", "bn", (co,)))
    return rows


def _chain(prefix, kind, dims, ksize, idx):
    """dims = [c0, c1, ...]; idx = Sequential positions of the weighted layers."""
    rows = []
    for (ci, co), i in zip(zip(dims[:-1], dims[1:]), idx):
        rows.append((f"{prefix}.{i}", kind, (co, ci) + tuple(ksize)))
    return rows


def detector_rows():
    rows = _vgg("backbone.features")
    rows += _chain("score_conv", "conv", [12, 64, 64], (1, 1, 1), (0, 2))
    for head, co in (("score_predict", 1), ("scale_predict", 1), ("offset_predict", 2)):
        rows += _chain(head, "conv", [64, 64, 64, co], (3, 3), (0, 2, 4))
    return rows


def selector_rows(an=5):
    rows = _vgg("backbone.features")
    k = (1, 3, 3)
    rows += _chain("corr_conv_list.0", "conv", [512, 64, 64, 128, 128, 256, 256], k, (1, 4, 7, 10, 13, 16))
    rows += _chain("corr_conv_list.1", "conv", [512, 128, 128, 256, 256], k, (1, 4, 7, 10))
    rows += _chain("corr_conv_list.2", "conv", [512, 256, 256], k, (1, 4))
    rows += _chain("corr_feats_conv", "conv", [768, 512, 512], (1, 1, 1), (0, 3))
    rows += _chain("score_process", "conv", [515, 512, 512], (1, 1), (0, 2))
    for i in range(2):
        for name in ("conv_key", "conv_query", "conv_feats", "conv_merge"):
            rows.append((f"atts.{i}.{name}", "conv", (512, 512, 1)))
        rows.append((f"atts.{i}.norm.norm", "ln", (512,)))
        rows += _chain(f"mlps.{i}", "conv", [1024, 512, 512], (1,), (0, 3))
    rows += _chain("score_predict", "conv", [512, 512, 1], (1,), (0, 2))
    rows += _chain("angle_predict", "conv", [515 * an, 512, 512, 1], (1,), (0, 2, 4))
    rows += _chain("view_point_encoder", "linear", [3, 128, 256, 512], (), (0, 2, 4))
    return rows


def refiner_rows():
    rows = _vgg("feature_net.backbone.features")
    k2, k3 = (3, 3), (3, 3, 3)
    rows += _chain("feature_net.conv0", "conv", [256, 64, 64], k2, (0, 3))
    rows += _chain("feature_net.conv1", "conv", [512, 256, 64], k2, (0, 3))
    rows += _chain("feature_net.conv2", "conv", [512, 256, 64], k2, (0, 3))
    rows += _chain("feature_net.conv_out", "conv", [192, 128, 128], k2, (0, 3))
    rows += _chain("volume_net.mean_embed", "conv", [256, 64, 64], k3, (0, 3))
    rows += _chain("volume_net.var_embed", "conv", [128, 64, 64], k3, (0, 3))
    for name, ci, co in (("conv0", 128, 64), ("conv1", 64, 128), ("conv2", 128, 128),
                         ("conv3", 128, 256), ("conv4", 256, 256)):
        rows.append((f"volume_net.{name}.0", "conv", (co, ci) + k3))
    rows += _chain("volume_net.conv5", "conv", [256, 512, 512], k3, (0, 3))
    rows.append(("regressor.fc.0.0", "linear", (512, 32768)))
    rows.append(("regressor.fc.1.0", "linear", (512, 512)))
    rows.append(("regressor.fcr", "linear", (4, 512)))
    rows.append(("regressor.fct", "linear", (2, 512)))
    rows.append(("regressor.fcs", "linear", (1, 512)))
    return rows


def expand(rows):
    """(prefix, kind, shape) rows -> ordered [(state_dict key, shape, role)]."""
    out = []
    for prefix, kind, shape in rows:
        if kind in ("conv", "linear"):
            out.append((prefix + ".weight", tuple(shape), "weight"))
            out.append((prefix + ".bias", (shape[0],), "bias"))
        elif kind == "bn":
            out.append((prefix + ".weight", shape, "gamma"))
            out.append((prefix + ".bias", shape, "beta"))
            out.append((prefix + ".running_mean", shape, "rmean"))
            out.append((prefix + ".running_var", shape, "rvar"))
            out.append((prefix + ".num_batches_tracked", (), "count"))
        elif kind == "ln":
            out.append((prefix + ".weight", shape, "gamma"))
            out.append((prefix + ".bias", shape, "beta"))
        else:
            raise ValueError(kind)
    return out


ROWS = {"detector": detector_rows, "selector": selector_rows, "refiner": refiner_rows}

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
