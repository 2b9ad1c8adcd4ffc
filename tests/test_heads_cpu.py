"""CPU: the parts of the decoder-head / MTL wrapper row that are plain torch modules - TamModule
(models/models.py:11-134) against a functional transcription of the reference's forward (F.conv2d / F.batch_norm /
F.conv_transpose2d on the module's own parameters), for every task count the reference defines."""
import pytest
import torch
import torch.nn.functional as F

from m3vit_amd.heads import TamModule


def _conv_bn(x, seq, training, stride=1, transposed=False):
    conv, bn = seq[0], seq[1]
    if transposed:
        y = F.conv_transpose2d(x, conv.weight, conv.bias, stride=stride, padding=1, output_padding=1)
    else:
        y = F.conv2d(x, conv.weight, conv.bias, stride=stride, padding=1)
    return F.batch_norm(y, None if training else bn.running_mean, None if training else bn.running_var, bn.weight, bn.bias,
                        training=training, eps=bn.eps)


def _reference_lines(m, feats, training):
    tasks = m.tasks
    f = [feats[t] for t in tasks]
    b, c, H, W = f[0].shape
    x = torch.stack(f, dim=1).reshape(b, len(tasks) * c, H, W)                            # :82-83
    B = torch.sigmoid(_conv_bn(F.relu(_conv_bn(x, m.layers0, training)), m.layers1, training))   # _block0 :55-60
    n = len(tasks)
    if n == 2:                                                                             # :87-95
        Fb = torch.cat((f[0] * B, f[1] * (1 - B)), dim=1)
    elif n == 3:
        Fb = torch.cat((f[0] * B, f[1] * (1 - B) / 2, f[2] * (1 - B) / 2), dim=1)
    elif n == 4:
        Fb = torch.cat((f[0] * B / 2, f[1] * B / 2, f[2] * (1 - B) / 2, f[3] * (1 - B) / 2), dim=1)
    else:
        Fb = torch.cat((f[0] * B / 2, f[1] * B / 2, f[2] * (1 - B) / 3, f[3] * (1 - B) / 3, f[4] * (1 - B) / 3), dim=1)
    Fb = F.relu(_conv_bn(Fb, m.layers2, training))                                         # _block2
    Fb = F.relu(_conv_bn(F.relu(_conv_bn(Fb, m.encoder0, training, stride=2)), m.encoder1, training, stride=2))
    M = torch.sigmoid(_conv_bn(F.relu(_conv_bn(Fb, m.decoder0, training, 2, True)), m.decoder1, training, 2, True))
    Ftam = torch.cat([v * (1 + M) for v in f], dim=1)                                      # :107-115
    out = {}
    for t in tasks:                                                                        # :119-127
        h = F.relu(_conv_bn(Ftam, m.layers3[t], training))
        out[t] = F.conv2d(h, m.layers4[t][0].weight, m.layers4[t][0].bias)
    return out


@pytest.mark.parametrize("n_tasks", [2, 3, 4, 5])
@pytest.mark.parametrize("training", [False, True])
def test_tam_module_matches_reference_lines(n_tasks, training):
    torch.manual_seed(n_tasks)
    tasks = ["semseg", "depth", "normals", "edge", "sal"][:n_tasks]
    nout = {"semseg": 5, "depth": 1, "normals": 3, "edge": 1, "sal": 2}
    m = TamModule(tasks, 8, nout).double()
    for mod in m.modules():                                        # non-trivial BN statistics / affine
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
    m.train(training)
    feats = {t: torch.randn(2, 8, 8, 12, dtype=torch.float64) for t in tasks}
    want = _reference_lines(m, feats, training)
    got = m(feats)
    assert list(got) == tasks
    for t in tasks:
        assert got[t].shape == (2, nout[t], 8, 12)
        assert torch.allclose(got[t], want[t], rtol=1e-10, atol=1e-12), t
    keys = set(m.state_dict())
    assert {"layers0.0.weight", "layers1.1.running_var", "encoder1.0.bias", "decoder0.0.weight",
            f"layers3.{tasks[0]}.0.weight", f"layers4.{tasks[-1]}.0.bias"} <= keys


def test_tam_module_rejects_unsupported_task_counts():
    with pytest.raises(ValueError):
        TamModule(["a"], 8, {"a": 1})
