"""Drop-in for the reference's SimCLR/Model_Util.py (:9-146) on MI355X: same
function names, argument dict keys and error behaviour; the optimisers are the
fused HIP update kernels (maai_hip.optim) behind the torch.optim interface."""
import math
import os
import shutil
import sys

import torch
import torch.nn as nn

try:
    import maai_hip  # noqa: F401
except ImportError:
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, ".."), os.path.join(_h, "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import optim as _optim


def _cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0):
    frac = min(global_step, decay_steps) / decay_steps
    return learning_rate * ((1 - alpha) * 0.5 * (1 + math.cos(math.pi * frac)) + alpha)


def _get_train_steps(num_examples, train_epochs, train_batch_size):
    return num_examples * train_epochs // train_batch_size + 1


def learning_rate_schedule(arguments):
    """Warm-up + cosine schedule written into every param group (Model_Util.py:9-39).
    The step is read from optimizer.state[<last param of group 0>]['step'] (1 if absent)."""
    opt = arguments['optimizer']
    st = opt.state[opt.param_groups[0]["params"][-1]]
    step = st['step'] if 'step' in st else 1
    bs = arguments['batch_size']
    warmup = int(round(arguments['warmup_epochs'] * arguments['num_examples'] // bs))
    gbs = arguments['world_size'] * bs
    rule = arguments['learning_rate_scaling']
    if rule == 'linear':
        scaled = arguments['base_learning_rate'] * gbs / 256.
    elif rule == 'sqrt':
        scaled = arguments['base_learning_rate'] * math.sqrt(gbs)
    else:
        raise ValueError('Unknown learning rate scaling {}'.format(rule))
    if step < warmup:
        lr = float(step) / int(warmup) * scaled
    else:
        total = _get_train_steps(arguments['num_examples'], arguments['train_epochs'], bs)
        lr = _cosine_decay(scaled, step - warmup, total - warmup)
    for group in opt.param_groups:
        group['lr'] = lr


def get_optimizer(model, args):
    """sgd / adam / 'lars' (= LARC around Adam, as the reference wires it; Model_Util.py:68-88)."""
    if args.optimizer == 'sgd':
        return _optim.HipSGD(model.parameters(), args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    if args.optimizer == 'adam':
        return _optim.HipAdam(model.parameters(), args.lr)
    if args.optimizer == 'lars':
        return _optim.LARC(_optim.HipAdam(model.parameters(), args.lr))
    raise ValueError('Unknown optimizer {}'.format(args.optimizer))


def save_checkpoint(state, is_best, filename='checkpoint.pth.tar', best_filename='model_best.pth.tar'):
    torch.save(state, filename)
    if is_best:
        print('Saving a new best model with precesion {}'.format(state['best_prec1']))
        shutil.copyfile(filename, best_filename)


def top_k_accuracy(preds, target, k):
    """fraction of rows whose target (class index, or argmax of a one-hot row) is among the top-k predictions"""
    top = torch.topk(preds, k=k, dim=1)[1]
    tgt = target if target.dim() == 1 else torch.argmax(target, dim=1)
    hit = (top == tgt.unsqueeze(1)).any(dim=1)
    return hit.sum() / (hit.shape[0] + 0.0)


class Identity(nn.Module):
    def forward(self, x):
        return x


def plot_features_stats(losses, top1_acc, top5_acc):
    import matplotlib.pyplot as plt  # the reference forgot this import (Model_Util.py:134)
    fig, axes = plt.subplots(3, 1, sharex=True, figsize=(10, 10))
    fig.suptitle('Training process history', fontweight="bold", size=20)
    for ax, series, label, colour in zip(axes, (losses, top1_acc, top5_acc),
                                         ('Loss', 'Top 1 contrastive accuracy', 'Top 5 contrastive accuracy'),
                                         ('tab:blue', 'tab:green', 'tab:orange')):
        ax.plot(series, colour)
        ax.set(ylabel=label)
    axes[-1].set(xlabel='Epochs')
    plt.show()
