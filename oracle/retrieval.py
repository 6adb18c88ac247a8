"""Oracle of the retrieval metric.  TEST INFRASTRUCTURE ONLY.

ref: tools/test_retrival.py:283-303, restated with numpy loops (no torch.topk): for every test feature, rank the train
features by cosine similarity after per-split centring and L2 normalisation; a hit at k = a same-label video among the k
most similar.  Pinned by tests/golden/retrieval_g10.json, produced by executing the reference's own source lines
(tools/oracle/make_golden_retrieval.py)."""
import numpy as np


def knn_accuracy(train_feature, train_label, test_feature, test_label, ks=(1, 5, 10, 20, 50)):
    tr = np.asarray(train_feature, dtype=np.float32)
    te = np.asarray(test_feature, dtype=np.float32)
    te = te - te.mean(axis=0, keepdims=True)
    tr = tr - tr.mean(axis=0, keepdims=True)
    te = te / np.maximum(np.linalg.norm(te, axis=1, keepdims=True), 1e-12)
    tr = tr / np.maximum(np.linalg.norm(tr, axis=1, keepdims=True), 1e-12)
    sim = te @ tr.T
    train_label, test_label = np.asarray(train_label), np.asarray(test_label)
    out = {}
    for k in ks:
        hits = 0
        for i in range(sim.shape[0]):
            nn = np.argsort(-sim[i], kind='stable')[:k]
            hits += int((train_label[nn] == test_label[i]).any())
        out[k] = hits / sim.shape[0]
    return out


def clustered_features(seed, n_train=300, n_test=120, classes=12, dim=512, noise=2.0):
    """seeded class-clustered features with a common offset (so that the centring step matters): the inputs of
    tests/golden/retrieval_g10.json"""
    import torch
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn((classes, dim), generator=g)
    tl = torch.randint(0, classes, (n_train,), generator=g)
    sl = torch.randint(0, classes, (n_test,), generator=g)
    tf = centres[tl] + noise * torch.randn((n_train, dim), generator=g) + 3.0
    sf = centres[sl] + noise * torch.randn((n_test, dim), generator=g) + 3.0
    return tf, tl, sf, sl
