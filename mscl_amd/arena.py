"""Flat parameter arenas: every trainable tensor of the model is a view into ONE fp32 buffer.

Why (MI355X-first): the parameter-sized passes of a step -- key-encoder EMA (recognizers/moco.py:408-421,
~300 tiny launches x3 in the reference), gradient-norm + clip + SGD (mmcv OptimizerHook, ~150 tensors),
the bf16 shadow refresh and the data-parallel gradient all-reduce -- each become ONE kernel / ONE
collective over a contiguous range, at HBM / xGMI speed, instead of hundreds of launches.

Layout (all arenas share offsets):
  Q   fp32  trainable parameters   [rgb: encoder_q | neck_q | mlp_q][flow: encoder_q | neck_q | mlp_q]
  KX  fp32  key-encoder twins      same order (encoder_k | neck_k | mlp_k)
  G   fp32  gradients of Q         (kernels accumulate straight into it; zeroed once per step)
  MOM fp32  SGD momentum buffers
  Qb/Kb bf16 shadows of Q / KX     (what the MFMA kernels read)
Conv kernels keep the logical shape (Cout,Cin,kT,kH,kW) of the reference's state dict but are stored
[Cout][kT][kH][kW][Cin] (channels_last_3d strides), which is exactly the implicit-GEMM B layout.
"""
import torch

ALIGN = 64          # elements; keeps every tensor 256-byte aligned in fp32 and 128-byte in bf16


def _physical_view(flat, off, shape):
    """view of flat[off: off+numel] with `shape` as LOGICAL shape; 5-D -> channels-last strides."""
    n = 1
    for s in shape:
        n *= s
    seg = flat[off:off + n]
    if len(shape) == 5:
        co, ci, kt, kh, kw = shape
        return seg.view(co, kt, kh, kw, ci).permute(0, 4, 1, 2, 3)
    return seg.view(shape)


class Slot:
    """one parameter's place in the arenas"""
    __slots__ = ('name', 'shape', 'off', 'numel', 'touched')

    def __init__(self, name, shape, off, numel):
        self.name, self.shape, self.off, self.numel, self.touched = name, tuple(shape), off, numel, False


class ParamArena:
    def __init__(self, device):
        self.device = torch.device(device)
        self.slots = []
        self.size = 0
        self.ranges = {}          # group name -> (start, end)

    def add(self, name, shape):
        n = 1
        for s in shape:
            n *= s
        slot = Slot(name, shape, self.size, n)
        self.slots.append(slot)
        self.size += (n + ALIGN - 1) // ALIGN * ALIGN
        return slot

    def begin_group(self, name):
        self.ranges[name] = [self.size, self.size]

    def end_group(self, name):
        self.ranges[name][1] = self.size

    def allocate(self, with_key=True, with_grad=True):
        dev, n = self.device, self.size
        self.Q = torch.zeros(n, dtype=torch.float32, device=dev)
        self.Qb = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        self.KX = torch.zeros(n, dtype=torch.float32, device=dev) if with_key else None
        self.Kb = torch.zeros(n, dtype=torch.bfloat16, device=dev) if with_key else None
        self.G = torch.zeros(n, dtype=torch.float32, device=dev) if with_grad else None
        self.MOM = torch.zeros(n, dtype=torch.float32, device=dev) if with_grad else None

    def view(self, which, slot):
        return _physical_view(getattr(self, which), slot.off, slot.shape)

    def packed(self, which, slot):
        """physical (contiguous) view: conv kernels as (Cout,kT,kH,kW,Cin)"""
        seg = getattr(self, which)[slot.off:slot.off + slot.numel]
        if len(slot.shape) == 5:
            co, ci, kt, kh, kw = slot.shape
            return seg.view(co, kt, kh, kw, ci)
        return seg.view(slot.shape)

    def active_ranges(self):
        """maximal contiguous [start, end) ranges covering the slots that ever received a gradient
        (SGD skips tensors whose grad is None: no weight decay, no momentum -- SURVEY.md App. E-10)."""
        out = []
        for s in self.slots:
            if not s.touched:
                continue
            a, b = s.off, s.off + (s.numel + ALIGN - 1) // ALIGN * ALIGN
            if out and out[-1][1] == a:
                out[-1][1] = b
            else:
                out.append([a, b])
        return [tuple(r) for r in out]
