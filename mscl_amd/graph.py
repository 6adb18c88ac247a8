"""Whole-step HIP-graph capture: forward + loss + backward + clip/SGD of MSCLWithAug as ONE graph launch.

Why: at ~900 kernel launches per step the eager step is host-bound (10.9 ms of Python + HIP launch calls per step on
the host for ~10 ms of three-stream GPU work on an MI355X; how fast the host is decides the eager rate).  Everything that changes from step to step is either device state (queues, counters, parameters),
an input copied into static buffers, or a per-step word (EMA momentum, learning rate, shuffle indices) that is uploaded
to a fixed device address BEFORE each replay, in stream order, through a ring of pinned slots (staging.py) -- so the
captured graph stays valid for the whole run and step n never sees the words of step n+k although the host runs ahead.
"""
import torch

from . import parallel


class GraphedStep:
    def __init__(self, model, optimizer, example_batch, warmup=2, indirect=None):
        """indirect: None = inputs by address wherever the step allows it (below), False = always through the static buffers"""
        self.model, self.opt = model, optimizer
        # The shuffle-BN exchange inside a captured step uses the all-gather formulation (W x the rows of the all-to-all).  Round 6 dealt
        # the permutation balanced (parallel.shuffle_perm: B / W rows to every rank, always), which gives the all-to-all equal, constant
        # split sizes -- and found that capturing `dist.all_to_all_single` segfaults in this stack (RCCL 2.26.6 under torch 2.10, one-rank
        # group, tools/scratch/a2a_capture.py; the all-gather captures and replays fine), so the fallback stays.
        model.shuffle_mode = 'gather'
        dev = model.arena.device
        example_batch = model.with_aug_draw(example_batch)      # stochastic augmenter: static mask / parameter buffers
        self.static = {k: [t.to(dev).clone() for t in v] for k, v in example_batch.items()}
        self.B = self.static[model.im_key][0].shape[0]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # eager warm-up: touched-parameter ranges, lazy kernel attributes
            for _ in range(warmup):
                self._eager()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # host-side state mutated by the (non-executing) capture pass is rolled back afterwards
        snap = self._host_state()
        parallel.settle_before_capture()
        self.graph = torch.cuda.CUDAGraph()
        fk = model.flow_key[0]
        model._pre_step_host(self.B)
        model._upload_step_words()
        optimizer.sync_lr()
        calls = [(g, g.calls) for g in list(model._key_graph) + list(model._query_graph)]
        # Inputs by ADDRESS where the step only reads them through mscl_pack_input (one GPU, deterministic augmentation, 3-channel
        # flow): the captured launches read each clip's address from a device word, a new batch costs a 32-byte upload instead of
        # a 116-MB copy into static buffers (~55 us of the step at B = 8, T = 16).  Everything else keeps the static copies.
        self.indirect = None
        ins = [self.static[model.im_key][0], self.static[model.im_key][1], self.static[fk][0], self.static[fk][1]]
        plain = (indirect is not False and parallel.single() and not getattr(model.aug_gpu, 'stochastic', False) and 'flip_mask' not in self.static
                 and all(t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == 3 for t in ins))
        if plain:
            from .kernels import IndirectInput
            from .staging import StagingRing
            self._ptr_ring = StagingRing((4,), torch.long, dev)
            self._ptr_cpu = torch.tensor([t.data_ptr() for t in ins], dtype=torch.long)
            self._ptr_ring.push(self._ptr_cpu)
            self.indirect = [IndirectInput(self._ptr_ring.dev[i:i + 1], t.shape, dev) for i, t in enumerate(ins)]
            self._ind_shapes = [tuple(t.shape) for t in ins]
            ins = self.indirect
        with torch.cuda.graph(self.graph):            # (capturing the main chain on a high-priority stream measured no gain)
            flips = self.static.get('flip_mask', (None, None))
            rows = self.static.get('aug_params', (None, None))
            loss, logs = model._device_step(ins[0], ins[1], ins[2], ins[3], flips[0], flips[1], rows[0], rows[1])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        self._restore(snap)
        for g, n in calls:          # the capture pass walked the sub-graph holders' eager branch: not a warm-up call of theirs
            g.calls = n
        self.loss, self.logs = loss.detach(), logs

    def _eager(self):
        out = self.model.train_step(self.static, sync_logs=False)
        self.opt.zero_grad()
        out['loss'].backward()
        self.opt.step()

    def _host_state(self):
        m = self.model
        return dict(step=m._step, steps=self.opt.steps,
                    rec=[(r.iters, r.batch_size, r.m) for r in (m.recognizer, m.recognizer_flow)])

    def _restore(self, s):
        m = self.model
        m._step, self.opt.steps = s['step'], s['steps']
        for r, (it, bs, mm) in zip((m.recognizer, m.recognizer_flow), s['rec']):
            r.iters, r.batch_size, r.m = it, bs, mm

    def step(self, batch):
        """one training step on `batch` (device tensors); returns (loss, logs) as static device tensors."""
        batch = self.model.with_aug_draw(batch)
        m = self.model
        srcs = [batch[m.im_key][0], batch[m.im_key][1], batch[m.flow_key[0]][0], batch[m.flow_key[0]][1]] if self.indirect else None
        if srcs is not None and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shp
                                    for t, shp in zip(srcs, self._ind_shapes)):
            for i, t in enumerate(srcs):
                self._ptr_cpu[i] = t.data_ptr()
            self._ptr_ring.push(self._ptr_cpu)
            self._live = (self._live[-2:] if hasattr(self, '_live') else []) + [srcs]      # the clips of the last replays stay referenced
            for k in ('flip_mask', 'aug_params'):   # what else the captured step READS travels through its static buffer; nothing
                if k in self.static and k in batch:  # else does ('label' would cost B scalar copy launches per replay for no reader)
                    for dst, src in zip(self.static[k], batch[k]):
                        if torch.is_tensor(dst) and dst.data_ptr() != src.data_ptr():
                            dst.copy_(src, non_blocking=True)
        else:
            if self.indirect:                       # a batch in another form: through the static buffers, addressed by the same words
                for i, (k, j) in enumerate(((m.im_key, 0), (m.im_key, 1), (m.flow_key[0], 0), (m.flow_key[0], 1))):
                    self._ptr_cpu[i] = self.static[k][j].data_ptr()
                self._ptr_ring.push(self._ptr_cpu)
            for k, v in self.static.items():
                for dst, src in zip(v, batch[k]):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src, non_blocking=True)
        self.model._pre_step_host(self.B)
        self.model._upload_step_words()
        self.opt.sync_lr()
        self.graph.replay()
        self.model._post_step_host()
        self.opt.steps += 1
        return self.loss, self.logs

    def log_vars(self):
        return self.model._parse_logs(self.logs, True)
