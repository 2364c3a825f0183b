"""Stand-ins for the reference's un-vendored dependencies (BUILD CONTAINER ONLY).

The reference hot path (``/root/reference/models/src/anemoi/models``) imports
``torch_geometric`` (>=2.3, unpinned: models/pyproject.toml:44), ``hydra``,
``omegaconf`` and ``anemoi.utils``; none of them is installed in this image and
there is no network.  This module injects minimal re-statements of the
*documented* semantics of the handful of symbols the hot path touches
 into ``sys.modules`` so that the reference's own Python
can be imported, *in this container only*, to generate the golden vectors under
``tests/golden/`` (see ``make_golden.py``).

Nothing here is shipped, imported by the product package, or executed on the GPU
box; it contains no reference source.
"""
from __future__ import annotations

import functools
import importlib
import inspect
import sys
import types
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

REF_ROOT = "/root/reference"


# --------------------------------------------------------------------------- torch_geometric
def _scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    """torch_geometric.utils.scatter: index along ``dim`` is 1-D and broadcast."""
    dim = dim if dim >= 0 else src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    if reduce in ("sum", "add"):
        return src.new_zeros(shape).index_add_(dim, index, src)
    if reduce in ("max", "amax"):
        view = [1] * src.dim()
        view[dim] = -1
        idx = index.view(view).expand_as(src)
        # PyG fills empty segments with 0 for 'max'
        return src.new_zeros(shape).scatter_reduce_(dim, idx, src, reduce="amax", include_self=False)
    raise NotImplementedError(reduce)


def _softmax(src: Tensor, index: Tensor, ptr=None, num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    """torch_geometric.utils.softmax: exp(x - segmax) / (segsum + 1e-16)."""
    n = num_nodes if num_nodes is not None else (int(index.max()) + 1 if index.numel() else 0)
    src_max = _scatter(src.detach(), index, dim, dim_size=n, reduce="max")
    out = (src - src_max.index_select(dim, index)).exp()
    out_sum = _scatter(out, index, dim, dim_size=n, reduce="sum") + 1e-16
    return out / out_sum.index_select(dim, index)


def _degree(index: Tensor, num_nodes: Optional[int] = None, dtype=None) -> Tensor:
    n = num_nodes if num_nodes is not None else int(index.max()) + 1
    out = torch.zeros(n, dtype=dtype or torch.get_default_dtype(), device=index.device)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype, device=index.device))


def _index_sort(inputs: Tensor, max_value: Optional[int] = None, stable: bool = False) -> Tuple[Tensor, Tensor]:
    return torch.sort(inputs, stable=True)


def _index2ptr(index: Tensor, size: Optional[int] = None) -> Tensor:
    if size is None:
        size = int(index.max()) + 1 if index.numel() > 0 else 0
    counts = torch.bincount(index, minlength=size)
    ptr = torch.zeros(size + 1, dtype=torch.long, device=index.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr


class _MessagePassing(torch.nn.Module):
    """Minimal torch_geometric.nn.conv.MessagePassing (flow source_to_target)."""

    def __init__(self, aggr: str = "add", node_dim: int = -2, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index: Tensor, size=None, **kwargs):
        size = list(size) if size is not None else [None, None]
        msg_params = list(inspect.signature(self.message).parameters)
        coll = {}
        for name in msg_params:
            if name.endswith("_i") or name.endswith("_j"):
                base, which = name[:-2], (1 if name.endswith("_i") else 0)
                if base == "size":
                    continue
                data = kwargs[base]
                if isinstance(data, (tuple, list)):
                    data = data[which]
                if isinstance(data, Tensor):
                    if size[which] is None:
                        size[which] = data.size(self.node_dim)
                    data = data.index_select(self.node_dim, edge_index[which])
                coll[name] = data
            elif name in kwargs:
                coll[name] = kwargs[name]
        if size[1] is None:
            size[1] = size[0]
        if size[0] is None:
            size[0] = size[1]
        special = {
            "index": edge_index[1],
            "ptr": None,
            "size_i": size[1],
            "size_j": size[0],
            "edge_index": edge_index,
        }
        for k, v in special.items():
            if k in msg_params:
                coll[k] = v
        out = self.message(**{k: coll[k] for k in msg_params})

        aggr_params = list(inspect.signature(self.aggregate).parameters)[1:]
        pool = dict(kwargs)
        pool.update(special)
        pool["dim_size"] = size[1]
        pool["index"] = edge_index[1]
        return self.aggregate(out, **{k: pool[k] for k in aggr_params if k in pool})

    def message(self, x_j):  # pragma: no cover
        return x_j

    def aggregate(self, inputs: Tensor, index: Tensor, dim_size: Optional[int] = None):
        return _scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce="sum")


class _Store(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            if k == "num_nodes" and "x" in self:
                return self["x"].shape[0]
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _HeteroData:
    """Tiny torch_geometric.data.HeteroData: stores auto-created on first access."""

    def __init__(self):
        object.__setattr__(self, "_nodes", {})
        object.__setattr__(self, "_edges", {})

    def __getitem__(self, key):
        table = self._edges if isinstance(key, tuple) else self._nodes
        if key not in table:
            table[key] = _Store()
        return table[key]

    def __bool__(self):
        return True

    @property
    def node_types(self):
        return list(self._nodes)

    @property
    def edge_types(self):
        return list(self._edges)

    def node_items(self):
        return list(self._nodes.items())

    def edge_items(self):
        return list(self._edges.items())


# --------------------------------------------------------------------------- PyG pickle layout (graph FILES only)
# What `torch.save(HeteroData)` puts on disk (torch-geometric >= 2.3, restated from its published source): the classes
# below have the same module / class names and the same __dict__ layout, so a file written through them unpickles like a
# file written by the real package.  They are used ONLY to write the graph-file fixture (make_golden.py: gen_edges).
class _PygBaseStorage:
    """torch_geometric.data.storage.BaseStorage: attributes live in ``_mapping``; ``_parent`` is pickled dereferenced."""

    def __init__(self, _parent=None, _key=None):
        self.__dict__["_mapping"] = {}
        if _parent is not None:
            self.__dict__["_parent"] = _parent
        if _key is not None:
            self.__dict__["_key"] = _key

    def __setattr__(self, k, v):
        if k.startswith("_"):
            self.__dict__[k] = v
        else:
            self.__dict__["_mapping"][k] = v


class _PygNodeStorage(_PygBaseStorage):
    pass


class _PygEdgeStorage(_PygBaseStorage):
    pass


class _PygDataTensorAttr:
    pass


class _PygDataEdgeAttr:
    pass


class _PygHeteroData:
    """torch_geometric.data.hetero_data.HeteroData: the five __dict__ entries a pickled instance carries."""

    def __init__(self):
        d = self.__dict__
        d["_edge_attr_cls"] = _PygDataEdgeAttr
        d["_tensor_attr_cls"] = _PygDataTensorAttr
        d["_global_store"] = _PygBaseStorage(_parent=self)
        d["_node_store_dict"] = {}
        d["_edge_store_dict"] = {}

    def __getitem__(self, key):
        if isinstance(key, tuple):
            table, cls = self.__dict__["_edge_store_dict"], _PygEdgeStorage
        else:
            table, cls = self.__dict__["_node_store_dict"], _PygNodeStorage
        if key not in table:
            table[key] = cls(_parent=self, _key=key)
        return table[key]


def _name_as(cls, module, name):
    cls.__module__, cls.__name__, cls.__qualname__ = module, name, name
    return cls


# --------------------------------------------------------------------------- hydra / omegaconf / anemoi.utils
class _DotDict(dict):
    """anemoi.utils.config.DotDict: dict with attribute access, recursive."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(i) for i in v]
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))


class _InstantiationException(Exception):
    pass


def _locate(path: str):
    mod, _, name = path.rpartition(".")
    try:
        return getattr(importlib.import_module(mod), name)
    except Exception as e:  # noqa: BLE001
        raise _InstantiationException(str(e)) from e


def _instantiate(config, *args, **kwargs):
    config = dict(config)
    kwargs = dict(kwargs)
    recursive = kwargs.pop("_recursive_", config.pop("_recursive_", True))
    partial = kwargs.pop("_partial_", config.pop("_partial_", False))
    config.pop("_convert_", None)
    kwargs.pop("_convert_", None)
    target = config.pop("_target_")
    cls = _locate(target) if isinstance(target, str) else target
    merged = {**config, **kwargs}
    if recursive:
        merged = {k: (_instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in merged.items()}
    if partial:
        return functools.partial(cls, *args, **merged)
    return cls(*args, **merged)


def install(extra_paths: Tuple[str, ...] = ()) -> None:
    """Inject the stand-ins and put the reference sources on sys.path."""
    if "torch_geometric" in sys.modules and getattr(sys.modules["torch_geometric"], "_standin", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        m._standin = True
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric")
    typing_m = mod("torch_geometric.typing")
    typing_m.Adj = Tensor
    typing_m.OptTensor = Optional[Tensor]
    typing_m.PairTensor = Tuple[Tensor, Tensor]
    typing_m.OptPairTensor = Tuple[Tensor, Optional[Tensor]]
    typing_m.Size = Optional[Tuple[int, int]]
    nn_m = mod("torch_geometric.nn")
    conv_m = mod("torch_geometric.nn.conv")
    conv_m.MessagePassing = _MessagePassing
    nn_m.conv = conv_m
    nn_m.MessagePassing = _MessagePassing
    utils_m = mod("torch_geometric.utils")
    utils_m.scatter = _scatter
    utils_m.softmax = _softmax
    utils_m.degree = _degree
    utils_m.index_sort = _index_sort
    sparse_m = mod("torch_geometric.utils.sparse")
    sparse_m.index2ptr = _index2ptr
    utils_m.sparse = sparse_m
    data_m = mod("torch_geometric.data")
    data_m.HeteroData = _HeteroData
    # the pickle-layout classes of graph FILES, under the module paths the real package pickles them with
    hd_m, st_m, dd_m = mod("torch_geometric.data.hetero_data"), mod("torch_geometric.data.storage"), mod("torch_geometric.data.data")
    hd_m.HeteroData = _name_as(_PygHeteroData, "torch_geometric.data.hetero_data", "HeteroData")
    st_m.BaseStorage = _name_as(_PygBaseStorage, "torch_geometric.data.storage", "BaseStorage")
    st_m.NodeStorage = _name_as(_PygNodeStorage, "torch_geometric.data.storage", "NodeStorage")
    st_m.EdgeStorage = _name_as(_PygEdgeStorage, "torch_geometric.data.storage", "EdgeStorage")
    dd_m.DataTensorAttr = _name_as(_PygDataTensorAttr, "torch_geometric.data.data", "DataTensorAttr")
    dd_m.DataEdgeAttr = _name_as(_PygDataEdgeAttr, "torch_geometric.data.data", "DataEdgeAttr")
    data_m.hetero_data, data_m.storage, data_m.data = hd_m, st_m, dd_m
    tg.typing, tg.nn, tg.utils, tg.data = typing_m, nn_m, utils_m, data_m

    hy = mod("hydra")
    hy_u = mod("hydra.utils")
    hy_u.instantiate = _instantiate
    hy_e = mod("hydra.errors")
    hy_e.InstantiationException = _InstantiationException
    hy.utils, hy.errors = hy_u, hy_e

    oc = mod("omegaconf")

    class DictConfig(_DotDict):
        pass

    class ListConfig(list):
        pass

    class OmegaConf:
        @staticmethod
        def create(x=None):
            return _DotDict(x or {})

        @staticmethod
        def to_container(x, resolve=True):
            return x

        @staticmethod
        def is_config(x):
            return isinstance(x, _DotDict)

    oc.DictConfig, oc.ListConfig, oc.OmegaConf = DictConfig, ListConfig, OmegaConf

    # anemoi namespace: real `anemoi.models` from the reference + stand-in `anemoi.utils`
    for p in (f"{REF_ROOT}/models/src", f"{REF_ROOT}/graphs/src", *extra_paths):
        if p not in sys.path:
            sys.path.insert(0, p)
    anemoi = types.ModuleType("anemoi")
    anemoi.__path__ = [f"{REF_ROOT}/models/src/anemoi", f"{REF_ROOT}/graphs/src/anemoi"]
    sys.modules["anemoi"] = anemoi
    au = mod("anemoi.utils")
    au.__path__ = []
    auc = mod("anemoi.utils.config")
    auc.DotDict = _DotDict
    au.config = auc
    anemoi.utils = au


DotDict = _DotDict
HeteroData = _HeteroData
PygHeteroData = _PygHeteroData
