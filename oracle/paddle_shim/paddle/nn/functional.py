import torch as _t


def sigmoid(x):
    return _t.sigmoid(x)


def softmax(x, axis=-1):
    return _t.softmax(x, dim=axis)


def log_loss(input, label, epsilon=1e-4):  # noqa: A002
    return -label * _t.log(input + epsilon) - (1 - label) * _t.log(1 - input + epsilon)


def binary_cross_entropy_with_logits(logit, label, reduction="mean"):
    return _t.nn.functional.binary_cross_entropy_with_logits(logit, label, reduction=reduction)


def cross_entropy(input, label, reduction="mean"):  # noqa: A002  softmax CE on raw scores, label [B,1] int64
    return _t.nn.functional.cross_entropy(input, label.reshape(-1), reduction=reduction)
