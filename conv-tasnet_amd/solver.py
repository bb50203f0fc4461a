"""Solver: epoch driver with the reference's constructor and observable behaviour (src/solver.py:13-221).

``Solver(data, model, optimizer, arg_solver)`` takes the same positional 13-tuple.  Behaviour kept on purpose,
quirks included:
  * reported epoch loss = sum / (n_batches + 1)                                   (src/solver.py:171,219-221)
  * LR halving compares with the PREVIOUS epoch; after three misses it halves on that and on every further
    non-improving epoch (the miss counter is only reset by an improvement); early stop at 7  (:114-133)
  * resume: weights, optimiser state, loss history; ``epochs`` grows by ``start_epoch + 1``   (:56-69)
  * best-validation model saved to save_folder/model_path, optional per-epoch checkpoints      (:94-101,135-146)

Underneath, one step (:188-196) is: HIP forward -> HIP PIT loss -> HIP backward -> [RCCL all-reduce of the
flat gradient when torch.distributed is initialised] -> fused clip + Adam kernel (FlatAdam) or, for any other
torch optimiser, clip_grad_norm_ + optimizer.step().  The model may be bare or anything exposing ``.module``.
visdom is optional and only imported when one of the visdom flags is set.

Data parallel (one process per GPU) keeps the reference's single-decision semantics of nn.DataParallel
(src/train.py:83-85 + src/solver.py:103-133): every step's loss is the mean over the GLOBAL minibatch (each rank's
gradient is weighted by its share of the utterances, so ragged shards are exact), the epoch train / validation losses
are all-reduced before the LR-halving / early-stop / best-model logic, so every rank takes the same decision at the same
epoch (no replica drift, no rank left alone in a collective), and only rank 0 prints and saves.
"""
import collections
import os
import time

import torch

from . import parallel
from .optim import FlatAdam
from .pit_criterion import cal_loss

SolverArgs = collections.namedtuple(
    "SolverArgs", "use_cuda epochs half_lr early_stop max_grad_norm save_folder enable_checkpoint continue_from "
                  "model_path print_freq visdom_enabled visdom_epoch visdom_id")


class _HalvingSchedule:
    """The reference's validation-driven LR policy, isolated so it can be unit-tested."""

    def __init__(self, enabled, early_stop):
        self.enabled, self.early_stop = bool(enabled), bool(early_stop)
        self.previous = float("inf")
        self.misses = 0

    def update(self, val_loss):
        """-> (halve_now, stop_now)"""
        halve = stop = False
        if self.enabled:
            if val_loss >= self.previous:
                self.misses += 1
                halve = self.misses >= 3
                stop = self.misses >= 7 and self.early_stop
            else:
                self.misses = 0
        self.previous = val_loss
        return halve, stop


class Solver(object):
    def __init__(self, data, model, optimizer, arg_solver):
        a = SolverArgs(*arg_solver)
        self.args = a
        self.tr_loader, self.cv_loader = data['tr_loader'], data['cv_loader']
        self.model, self.optimizer = model, optimizer
        # attribute names the reference exposes
        self.use_cuda, self.epochs, self.half_lr, self.early_stop = a.use_cuda, a.epochs, a.half_lr, a.early_stop
        self.max_norm, self.save_folder, self.enable_checkpoint = a.max_grad_norm, a.save_folder, a.enable_checkpoint
        self.continue_from, self.model_path, self.print_freq = a.continue_from, a.model_path, a.print_freq
        self.visdom_enabled, self.visdom_epoch, self.visdom_id = a.visdom_enabled, a.visdom_epoch, a.visdom_id
        self.tr_loss = torch.Tensor(self.epochs)      # uninitialised history buffers, as in the reference
        self.cv_loss = torch.Tensor(self.epochs)
        self.iter_losses = []                         # every batch loss seen, python floats
        self._plot = self._make_plotter() if (a.visdom_enabled or a.visdom_epoch) else None
        self._rank0 = (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0
        if isinstance(optimizer, FlatAdam):
            # N > 1: the gradient all-reduce goes out in one bucket per repeat while the backward pass is still running
            parallel.enable_overlap(optimizer, int(getattr(self.net, "X", 0) or 0))
        self._reset()

    # -- plumbing -----------------------------------------------------------------------------------
    @property
    def net(self):
        """The ConvTasNet itself, whether or not it is wrapped (the reference insists on .module)."""
        return getattr(self.model, 'module', self.model)

    def _make_plotter(self):
        try:
            from visdom import Visdom
        except ImportError as e:
            raise RuntimeError("visdom plotting requested but the visdom package is not installed") from e
        vis, state = Visdom(env=self.visdom_id), {"win": None}

        def plot(epoch):
            xs = torch.arange(1, epoch + 2)
            ys = torch.stack((self.tr_loss[:epoch + 1], self.cv_loss[:epoch + 1]), dim=1)
            opts = dict(title=self.visdom_id, ylabel='Loss', xlabel='Epoch', legend=['train loss', 'cv loss'])
            if state["win"] is None:
                state["win"] = vis.line(X=xs, Y=ys, opts=opts)
            else:
                vis.line(X=xs.unsqueeze(0).expand(ys.size(1), xs.size(0)).transpose(0, 1), Y=ys, win=state["win"],
                         update='replace')
        return plot

    def _reset(self):
        self.start_epoch = 0
        if self.continue_from:
            self._say('Loading checkpoint model %s' % self.continue_from)
            pkg = torch.load(self.continue_from, map_location='cpu', weights_only=True)   # tensors / numbers / strings only
            self.net.load_state_dict(pkg['state_dict'])
            self.optimizer.load_state_dict(pkg['optim_dict'])
            self.start_epoch = int(pkg.get('epoch', 1))
            self.epochs = self.epochs + self.start_epoch + 1
            done = self.start_epoch
            self.tr_loss, self.cv_loss = torch.Tensor(self.epochs), torch.Tensor(self.epochs)
            self.tr_loss[:done] = pkg['tr_loss'][:done]
            self.cv_loss[:done] = pkg['cv_loss'][:done]
        os.makedirs(self.save_folder, exist_ok=True)
        self.schedule = _HalvingSchedule(self.half_lr, self.early_stop)
        self.best_val_loss = float("inf")

    # reference attribute names for the schedule state
    @property
    def prev_val_loss(self):
        return self.schedule.previous

    @property
    def val_no_impv(self):
        return self.schedule.misses

    def _save(self, path, epoch):
        if self._rank0:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            net = self.net
            torch.save(net.serialize(net, self.optimizer, epoch, tr_loss=self.tr_loss, cv_loss=self.cv_loss), path)

    def _say(self, msg):
        if self._rank0:
            print(msg, flush=True)

    def _halve_lr(self):
        sd = self.optimizer.state_dict()            # round trip through state_dict, like the reference
        sd['param_groups'][0]['lr'] = sd['param_groups'][0]['lr'] / 2.0
        self.optimizer.load_state_dict(sd)
        self._say('Learning rate adjusted to: %.6f' % sd['param_groups'][0]['lr'])

    # -- the loop -------------------------------------------------------------------------------------
    def train(self):
        bar = '-' * 85
        for epoch in range(self.start_epoch, self.epochs):
            t0 = time.time()
            self.model.train()
            tr = self._run_one_epoch(epoch)
            self._say('%s\nTrain Summary | End of Epoch %d | Time %.2fs | Train Loss %.3f\n%s'
                      % (bar, epoch + 1, time.time() - t0, tr, bar))
            if self.enable_checkpoint:
                path = os.path.join(self.save_folder, "checkpoint_models", 'epoch%d.pth.tar' % (epoch + 1))
                self._save(path, epoch + 1)
                self._say('Saving checkpoint model to %s' % path)

            self.model.eval()
            cv = self._run_one_epoch(epoch, cross_valid=True)
            self._say('%s\nValid Summary | End of Epoch %d | Time %.2fs | Valid Loss %.3f\n%s'
                      % (bar, epoch + 1, time.time() - t0, cv, bar))

            halve, stop = self.schedule.update(cv)
            if stop:
                self._say("No improvement for 7 epochs, early stopping.")
                break
            if halve:
                self._halve_lr()

            self.tr_loss[epoch], self.cv_loss[epoch] = tr, cv
            if cv < self.best_val_loss:
                self.best_val_loss = cv
                path = os.path.join(self.save_folder, self.model_path)
                self._save(path, epoch + 1)
                self._say("Found better validated model, saving to %s" % path)
            if self._plot is not None and self.visdom_enabled:
                self._plot(epoch)

    def _global_loss(self, loss, n_local):
        """(loss to back-propagate, loss to report) of the GLOBAL minibatch: with world ranks the reported loss is
        sum_r n_r loss_r / sum_r n_r, and the back-propagated one is scaled so that the 1/world average of the gradient
        all-reduce reproduces exactly that mean (n_r = utterances on rank r; equal shards -> scale 1)."""
        world = parallel.world_size()
        if world == 1:
            return loss, loss
        import torch.distributed as dist
        t = torch.stack((loss.detach() * n_local, loss.detach().new_tensor(float(n_local))))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return loss * (n_local * world / t[1]), t[0] / t[1]

    def _optimise(self, loss):
        """zero_grad -> backward -> (all-reduce) -> clip -> step."""
        opt = self.optimizer
        opt.zero_grad()
        loss.backward()
        if isinstance(opt, FlatAdam):
            opt.step(max_grad_norm=self.max_norm, grad_scale=parallel.allreduce_gradients(opt))
            return
        params = list(self.model.parameters())
        parallel.allreduce_gradients(params)
        torch.nn.utils.clip_grad_norm_(params, self.max_norm)
        opt.step()

    def _run_one_epoch(self, epoch, cross_valid=False):
        loader = self.cv_loader if cross_valid else self.tr_loader
        if not cross_valid and hasattr(getattr(loader, "dataset", None), "set_epoch"):
            loader.dataset.set_epoch(epoch)         # N > 1: another deal of the minibatches over the ranks every epoch
        dev = next(self.model.parameters()).device
        world = parallel.world_size()
        t0, running, n = time.time(), 0.0, 0
        for mixture, lengths, sources in loader:
            mixture, lengths, sources = mixture.to(dev), lengths.to(dev), sources.to(dev)
            with torch.set_grad_enabled(not cross_valid):
                loss = cal_loss(sources, self.model(mixture), lengths)[0]
            if not cross_valid:
                # training steps are collective (equal step counts per rank: data.AudioDataset equalises the plan)
                loss, report = self._global_loss(loss, int(mixture.shape[0]))
                self._optimise(loss)
            else:
                report = loss
            value = report.item()
            self.iter_losses.append(value)
            running += value
            if n % self.print_freq == 0 and self._rank0:
                print('Epoch %d | Iter %d | Average Loss %.3f | Current Loss %.6f | %.1f ms/batch'
                      % (epoch + 1, n + 1, running / (n + 1), value, 1000 * (time.time() - t0) / (n + 1)), flush=True)
            n += 1
        if cross_valid and world > 1:
            # validation minibatches are dealt to the ranks (any counts): sum and count over ALL of them give the number the
            # single-process reference computes, and every rank gets the same value for the schedule
            import torch.distributed as dist
            t = torch.tensor([running, float(n)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            running, n = float(t[0]), int(round(float(t[1])))
        return running / (n + 1)        # (n_batches + 1): the reference's divisor, kept for trajectory parity
