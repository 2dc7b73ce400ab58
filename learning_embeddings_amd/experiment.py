"""Host mirror of network/experiment.py (Experiment :23-224) and of the classifier trainer on config 4's path
(finetuner.CIFAR10.pass_samples :199-246, ethec_experiments.ETHECExperiment :202-240): model forward -> criterion ->
backward -> optimizer step, with the CNN parameters in one flat arena (single all-reduce, single Adam launch)."""
import copy
import os
import time

import torch
import torch.nn as nn

from . import parallel
from .resnet import resnet18, resnet50


class _NullWriter:
    def add_scalar(self, *a, **k): pass
    def close(self): pass


class Experiment:
    """experiment.py:23-224: epoch loop, MultiStepLR, checkpoints.  `evaluator` may be None (host-side reporting is
    out of scope); `pass_samples` is the base (inputs, labels) variant."""

    def __init__(self, model, dataloaders, criterion, classes, experiment_name, n_epochs, eval_interval, batch_size,
                 exp_dir, load_wt, evaluator, lr_step=[]):
        self.epoch = 0
        self.exp_dir = exp_dir; self.load_wt = load_wt; self.eval = evaluator
        self.classes = classes; self.criterion = criterion; self.batch_size = batch_size
        if not torch.cuda.is_available():
            raise RuntimeError('Experiment runs on the MI355X only (no CPU fallback)')
        self.device = torch.device('cuda', torch.cuda.current_device())
        print('Using device: {}'.format(self.device))
        self.model = model.to(self.device)
        self.n_epochs = n_epochs; self.eval_interval = eval_interval; self.dataloaders = dataloaders
        self.log_dir = os.path.join(self.exp_dir, '{}').format(experiment_name)
        self.path_to_save_model = os.path.join(self.log_dir, 'weights')
        self.make_dir_if_non_existent(self.path_to_save_model)
        self.writer = _NullWriter()
        self.lr_step = lr_step
        self.best_score = 0.0; self.best_model_wts = None

    @staticmethod
    def make_dir_if_non_existent(dir):
        if not os.path.exists(dir):
            os.makedirs(dir, exist_ok=True)

    def set_parameter_requires_grad(self, feature_extracting):
        if feature_extracting:
            for param in self.model.parameters():
                param.requires_grad = False

    def pass_samples(self, phase, save_to_tensorboard=True):
        running_loss = torch.zeros((), device=self.device); n = 0
        self.model.train(phase == 'train')
        for inputs, labels in self.dataloaders[phase]:
            inputs = inputs.to(self.device); labels = labels.to(self.device)
            self.optimizer.zero_grad()
            with torch.set_grad_enabled(phase == 'train'):
                outputs = self.model(inputs)
                loss = self.criterion(outputs, labels)
                if phase == 'train':
                    loss.backward(); self.optimizer.step()
            running_loss += loss.detach() * inputs.size(0); n += inputs.size(0)
        epoch_loss = running_loss.item() / max(n, 1)
        print('{} Loss: {:.4f}'.format(phase, epoch_loss))
        return epoch_loss

    def run_model(self, optimizer):
        self.optimizer = optimizer
        scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=self.lr_step, gamma=0.1)
        if self.load_wt:
            self.find_existing_weights()
        self.best_model_wts = copy.deepcopy(self.model.state_dict()); self.best_score = 0.0
        since = time.time()
        for self.epoch in range(self.epoch, self.n_epochs):
            print('=' * 10); print('Epoch {}/{}'.format(self.epoch, self.n_epochs - 1)); print('=' * 10)
            self.pass_samples(phase='train')
            if self.epoch % self.eval_interval == 0:
                self.pass_samples(phase='val'); self.pass_samples(phase='test')
            scheduler.step()
        print('Training complete in {:.0f}s'.format(time.time() - since))
        self.writer.close()
        return self.model

    def save_model(self, loss, filename=None):
        torch.save({'epoch': self.epoch, 'model_state_dict': self.model.state_dict(),
                    'optimizer_state_dict': self.optimizer.state_dict(), 'loss': loss},
                   os.path.join(self.path_to_save_model, '{}.pth'.format(filename if filename else self.epoch)))

    def load_model(self, epoch_to_load):
        ck = torch.load(os.path.join(self.path_to_save_model, '{}.pth'.format(epoch_to_load)), map_location=self.device)
        self.model.load_state_dict(ck['model_state_dict']); self.model = self.model.to(self.device)
        self.optimizer.load_state_dict(ck['optimizer_state_dict']); self.epoch = ck['epoch']

    def find_existing_weights(self):
        weights = sorted(os.listdir(self.path_to_save_model))
        if len(weights) < 2:
            print('Could not find weights to load from, will train from scratch.')
        else:
            self.load_model(epoch_to_load=weights[-2].split('.')[0])


class ETHECExperiment(Experiment):
    """Config 4's trainer: finetuner.CIFAR10 (model zoo pick :121-122, fc swap :150-157, pass_samples :199-246) as used
    by ethec_experiments.ETHECExperiment (:202-240).  criterion(outputs, labels, level_labels) -> scalar."""

    def __init__(self, data_loaders, labelmap, criterion, lr, batch_size=8, evaluator=None, experiment_name='exp',
                 experiment_dir='../exp/', n_epochs=10, eval_interval=2, feature_extracting=False, use_pretrained=False,
                 load_wt=False, model_name='resnet50', optimizer_method='adam', compute_dtype=torch.float32, weights=None, fast_path=True):
        self.labelmap = labelmap; self.lr = lr; self.model_name = model_name
        self.n_classes = labelmap.n_classes; self.levels = labelmap.levels; self.n_levels = len(labelmap.levels)
        self.rank, self.local_rank, self.world = parallel.init_process_group()
        if self.world > 1:
            torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
        model = {'resnet18': resnet18, 'resnet50': resnet50}[model_name]()
        if weights is not None:
            model.load_state_dict(weights)
        model.fc = nn.Linear(model.fc.in_features, self.n_classes)          # finetuner.py:150-157
        model = model.to(memory_format=torch.channels_last)
        Experiment.__init__(self, model, data_loaders, criterion, labelmap.classes, experiment_name, n_epochs,
                            eval_interval, batch_size, experiment_dir, load_wt, evaluator)
        self.compute_dtype = compute_dtype
        self.arena = parallel.FlatArena(self.model.parameters(), self.device)
        self.reducer = parallel.GradientReducer(self.arena)
        if self.world > 1:
            torch.distributed.broadcast(self.arena.data, 0)
        # liblecone's convolutions / fused BatchNorm / arena gradients / side-stream weight gradients, as in engine.StepEngine
        from .resnet import WgradOverlap
        self.overlap = None
        if fast_path and compute_dtype in (torch.float32, torch.bfloat16):
            if compute_dtype == torch.bfloat16:
                self.arena.enable_lowp_transposed()     # bf16 shadow + its transposed twin (the data gradients' operand)
            self.overlap = WgradOverlap(self.reducer, self.arena, side_stream=True)
        self.model.wgrad_overlap = self.overlap if self.overlap is not None else False    # this trainer's own (resnet.ResNet.wgrad_overlap)

    def fwd_bwd(self, inputs, level_labels, labels=None):
        """Forward, criterion, backward of one batch already on the device (NHWC): the region a hipGraph can capture."""
        self.arena.zero_grad()
        with torch.autocast('cuda', dtype=self.compute_dtype, enabled=self.compute_dtype != torch.float32):
            outputs = self.model(inputs)
        loss = self.criterion(outputs.float(), labels, level_labels)
        loss.backward()
        if self.overlap is not None:
            self.overlap.join()
        # detached: a live `outputs` would keep this step's autograd graph (and its AccumulateGrad nodes, bound to this step's
        # stream) alive into the next one -- under hipGraph capture that ends in a crash inside capture_end
        return loss.detach(), outputs.detach()

    def train_step(self, inputs, labels, level_labels):
        """finetuner.py:213-246 for one batch: forward, criterion, backward, (SUM all-reduce), Adam.  The criterion's
        mean is over the LOCAL batch, so under DP the summed gradient is divided by the world size."""
        pend = self.__dict__.setdefault('_steps_in_flight', [])     # host at most two steps ahead (see JointEmbeddings.train_step)
        if len(pend) >= 2:
            pend.pop(0).synchronize()
        inputs = inputs.to(self.device, non_blocking=True).contiguous(memory_format=torch.channels_last)
        loss, outputs = self.fwd_bwd(inputs, level_labels.to(self.device), labels)
        self.reducer.finish()
        self.arena.adam_step(self.lr, grad_scale=1.0 / self.world)
        if loss.is_cuda:
            done = torch.cuda.Event(); done.record(); pend.append(done)
        return loss, outputs
