"""Host-side helpers with the reference's ``utils`` surface (/root/reference/utils.py:10-74):
checkpoint save/load in the reference's ``{"epoch", "model"}`` format, LossDict, SingletonWriter."""
import os
import pickle

import torch


def load_model(model, pretrained, device):
    """utils.py:10-12; weights_only load (the file holds tensors and an int)."""
    state = torch.load(pretrained, map_location=device, weights_only=True)
    model.load_state_dict(state["model"], strict=False)


def save_losses(fig_dir, kls_real, kls_fake, kls_rec, rec_errs):
    """utils.py:15-23."""
    payload = dict(kl_real=kls_real, kl_fake=kls_fake, kl_rec=kls_rec, rec_err=rec_errs)
    with open(os.path.join(fig_dir, "soft_intro_train_graphs_data.pickle"), "wb") as fp:
        pickle.dump(payload, fp)


def save_checkpoint(model, epoch, iteration, prefix=""):
    """utils.py:26-36: ./saves/<prefix>model_epoch_{e}_iter_{i}.pth = {"epoch", "model": state_dict}."""
    os.makedirs("./saves/", exist_ok=True)
    path = "./saves/" + prefix + "model_epoch_{}_iter_{}.pth".format(epoch, iteration)
    torch.save({"epoch": epoch, "model": model.state_dict()}, path)
    print("model checkpoint saved @ {}".format(path))


def check_non_finite_gradints(model):
    """utils.py:39-45."""
    for name, param in model.named_parameters():
        if param.grad is not None:
            bad = (~torch.isfinite(param.grad)).sum().item()
            if bad:
                print("Non-finite gradients in ", name, bad, "values")


class LossDict(dict):
    """utils.py:48-60: key-wise + and scalar /."""

    def __add__(self, other):
        return LossDict({k: self.get(k, 0) + other.get(k, 0) for k in sorted(set(self) | set(other))})

    def __truediv__(self, value):
        return LossDict({k: v / value for k, v in self.items()})


class SingletonWriter(object):
    """utils.py:62-74: process-wide holder of the TensorBoard writer and the iteration counter."""
    writer = None
    cur_iter = 0
    test_iter = 1

    def __new__(cls):
        if not hasattr(cls, "instance"):
            cls.instance = super().__new__(cls)
        return cls.instance

    @property
    def write_test_iter(self):
        return self.writer and self.cur_iter % self.test_iter == 0
