"""Nearest-neighbour video retrieval on extracted features.

ref: tools/test_retrival.py:258-303 -- features of the train and the test split from `Recognizer3D` with
test_cfg['feature_extraction'] = True (one (512,) vector per video: the trunk's pooled layer-4 map, clips kept apart by
the dataset), then: centre each split on its own mean, L2-normalise, cosine similarity test x train, and for
k in (1, 5, 10, 20, 50) the share of test videos with a same-label video among their k nearest train videos.

The heavy part -- the trunk forward in evaluation mode -- runs on the HIP kernels (`Recognizer3D.feature_extraction`);
the metric itself is a (N_test, N_train) product of 512-vectors and a top-k, done with torch on the features' device.
"""
import torch
import torch.nn.functional as F

KS = (1, 5, 10, 20, 50)


@torch.no_grad()
def extract_features(model, batches):
    """features of every video of `batches` (iterable of dicts with 'imgs' (N, clips, 3, T, H, W) device tensors), in order.
    The model must have been built with test_cfg=dict(feature_extraction=True) (recognizers/base.py:103-106); it is run in
    evaluation mode (tools/test_retrival.py:160-177 `single_gpu_test`) and put back afterwards."""
    if not getattr(model, 'feature_extraction', False):
        raise ValueError('retrieval needs a recognizer built with test_cfg=dict(feature_extraction=True)')
    was = model.training
    model.eval()
    try:
        feats = [model._do_test(b['imgs']) for b in batches]
    finally:
        model.train(was)
    return torch.cat(feats).float()


@torch.no_grad()
def knn_accuracy(train_feature, train_label, test_feature, test_label, ks=KS):
    """ref: tools/test_retrival.py:283-303.  Returns {k: accuracy}."""
    if len(train_feature) != len(train_label) or len(test_feature) != len(test_label):
        raise AssertionError(f'{len(train_feature)} vs {len(train_label)}, {len(test_feature)} vs {len(test_label)}')
    train_label = torch.as_tensor(train_label, device=train_feature.device)
    test_label = torch.as_tensor(test_label, device=test_feature.device)
    test_feature = test_feature - test_feature.mean(dim=0, keepdim=True)
    train_feature = train_feature - train_feature.mean(dim=0, keepdim=True)
    test_feature = F.normalize(test_feature, p=2, dim=1)
    train_feature = F.normalize(train_feature, p=2, dim=1)
    sim = test_feature.matmul(train_feature.t())
    out = {}
    for k in ks:
        idx = torch.topk(sim, k, dim=1).indices
        out[k] = torch.any(train_label[idx] == test_label.unsqueeze(1), dim=1).float().mean().item()
    return out


def retrieval(model, train_batches, train_label, test_batches, test_label, ks=KS):
    """the whole of tools/test_retrival.py:258-303 on a materialised Recognizer3D"""
    return knn_accuracy(extract_features(model, train_batches), train_label, extract_features(model, test_batches), test_label, ks)
