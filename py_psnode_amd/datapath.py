"""Data path in front of the integrator: npz dataset -> pinned host staging -> HBM, once;
batches are then formed ON the device.

The scripts rebuild every batch on the host: `DataLoader` default-collates B samples of [T,D] into fresh [B,T,D] CPU
tensors and the loop moves each of the six (ODE) / nine (DAE) tensors with a pageable `.to(device)`
(neural_00_ODE_01_no_encode.py:288,343-347).  At the north-star batch that is ~0.3 GB of host copies + PCIe per step for a
forward that takes 4.9 ms.  These datasets are tiny next to 288 GB of HBM, so the MI355X-native data path keeps the
whole `ODE_Curves_Sample` / `DAE_Curves_Sample` (neural_base.py:10-40,136-166) resident and a batch is one
`index_select` per tensor: B-major contiguous [B,T,D], i.e. exactly the memory layout (and therefore the strided
`permute(1,0,2)` views) the collate + `.to(device)` route hands the solver.

`ResidentLoader` draws its permutation the way `DataLoader(shuffle=True)`'s RandomSampler does (a seed from the global
torch RNG, `randperm` on a fresh generator), so under the same RNG state it yields the same batches in the same order
as the scripts' loader (tests/test_datapath.py)."""
from typing import Iterator, Optional, Sequence, Tuple

import torch

ODE_FIELDS = ("t", "x", "z", "event_t", "z_jump", "mask")
DAE_FIELDS = ("t", "x", "z", "v", "i", "event_t", "z_jump", "v_jump", "mask")


def dataset_fields(dataset) -> Tuple[str, ...]:
    """Field order of `dataset.__getitem__` (neural_base.py:39-40,165-166): DAE datasets carry v, i, v_jump."""
    return DAE_FIELDS if all(hasattr(dataset, k) for k in ("v", "i", "v_jump")) else ODE_FIELDS


class ResidentDataset:
    """All tensors of an ODE_/DAE_Curves_Sample (ours or the reference's: duck-typed) staged once into device memory
    through pinned host buffers.  fp32 is required, as in the scripts ("dtype must already be float32", App. C)."""

    def __init__(self, dataset, device, fields: Optional[Sequence[str]] = None):
        self.device = torch.device(device)
        self.fields = tuple(fields) if fields is not None else dataset_fields(dataset)
        self.data_name = getattr(dataset, "data_name", None)
        n = None
        self.tensors = {}
        staged = []
        for key in self.fields:
            src = getattr(dataset, key)
            if not torch.is_tensor(src):
                raise TypeError(f"dataset.{key} is not a tensor")
            if src.dtype != torch.float32:
                raise TypeError(f"dataset.{key} is {src.dtype}; the path is fp32-only")
            if n is None:
                n = src.shape[0]
            elif src.shape[0] != n:
                raise ValueError(f"dataset.{key} has {src.shape[0]} samples, expected {n}")
            src = src.contiguous()
            if self.device.type == "cuda" and src.device.type == "cpu":
                src = src.pin_memory()          # one DMA-able staging buffer per field; async copy below
                staged.append(src)
                self.tensors[key] = src.to(self.device, non_blocking=True)
            else:
                self.tensors[key] = src.to(self.device)
        self.n = int(n or 0)
        if staged:
            torch.cuda.current_stream(self.device).synchronize()   # staging buffers may be freed after this

    def __len__(self):
        return self.n

    def __getattr__(self, key):
        tensors = self.__dict__.get("tensors", {})
        if key in tensors:
            return tensors[key]
        raise AttributeError(key)

    def batch(self, index: torch.Tensor):
        """The batch of samples `index` (int64, on the device): tuple in __getitem__ order, each B-major contiguous."""
        return tuple(self.tensors[k].index_select(0, index) for k in self.fields)

    def nbytes(self) -> int:
        return sum(v.numel() * v.element_size() for v in self.tensors.values())


class ResidentLoader:
    """Drop-in for the scripts' `DataLoader(dataset, batch_size=..., shuffle=...)` over a ResidentDataset: same batch
    order under the same torch RNG state, batches already on the device."""

    def __init__(self, resident: ResidentDataset, batch_size: int = 1, shuffle: bool = False, drop_last: bool = False,
                 generator: Optional[torch.Generator] = None):
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = resident, int(batch_size), shuffle, drop_last, generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self) -> torch.Tensor:
        n = len(self.dataset)
        if not self.shuffle:
            return torch.arange(n)
        gen = self.generator
        if gen is None:
            # what DataLoader(shuffle=True) consumes from the global RNG per epoch: the iterator's base seed
            # (_BaseDataLoaderIter.__init__), then RandomSampler.__iter__'s seed for a fresh generator
            torch.empty((), dtype=torch.int64).random_()
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
        return torch.randperm(n, generator=gen)

    def __iter__(self) -> Iterator[tuple]:
        order = self._order().to(self.dataset.device, non_blocking=True)
        n = order.numel()
        for lo in range(0, n, self.batch_size):
            idx = order[lo:lo + self.batch_size]
            if self.drop_last and idx.numel() < self.batch_size:
                return
            yield self.dataset.batch(idx)
