"""Reading the reference's on-disk graph — scope row f4.

``anemoi-graphs`` writes its graph with ``torch.save(HeteroData)`` and ``anemoi-training`` reads it back with
``torch.load(graph_filename, map_location=..., weights_only=False)``
(reference graphs/src/anemoi/graphs/create.py:179-189, training/src/anemoi/training/train/train.py:190-262).  The pickle
therefore names classes of the un-vendored third-party dependency ``torch_geometric`` (>= 2.3, unpinned:
models/pyproject.toml:44): ``torch_geometric.data.hetero_data.HeteroData``, the storages of
``torch_geometric.data.storage`` and the attribute classes of ``torch_geometric.data.data``.  That package is not part of
this stack, and it is not needed to READ the file: ``load_graph_from_file`` unpickles with a RESTRICTED unpickler that

 * rebuilds tensors / numpy arrays / builtin containers as usual,
 * maps every ``torch_geometric.*`` class to an inert placeholder that just keeps the pickled state, and
 * refuses everything else (a graph file is data; arbitrary globals are not executed),

and then walks PyG's published state layout (``HeteroData.__dict__`` = ``_global_store``, ``_node_store_dict`` {name:
NodeStorage}, ``_edge_store_dict`` {(src, relation, dst): EdgeStorage}; ``BaseStorage.__dict__`` = ``_mapping`` {attribute:
value}, ``_parent``, ``_key``) into a plain ``GraphData`` of dict-like stores.  ``GraphData`` offers the slice of the
HeteroData interface the model glue reads (``node_types``, ``edge_types``, ``g[name].x``, ``g[(src, "to", dst)].edge_index``,
attribute and item access), so it can be handed to ``AnemoiModelEncProcDec(graph_data=...)`` and ``create_graph_provider``
directly.

Parity note: the layout is restated from torch-geometric's published source, not checked against an installed copy (there
is none in the build container); the unpickler is tolerant by construction (any ``torch_geometric`` class name, any extra
keys) and the fixture ``tests/golden/graph_file.pt`` is written through stand-in classes with exactly that layout."""
from __future__ import annotations

import pickle
from typing import Any

import torch

_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "complex"), ("builtins", "bytearray"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "object"),
    ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"), ("_codecs", "encode"),  # protocol-2 pickles spell bytes that way
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"),
    ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"), ("torch.storage", "UntypedStorage"),
    ("torch.storage", "TypedStorage"), ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"),
}


class _Placeholder:
    """Stands in for any ``torch_geometric`` class named by the pickle: constructible with any arguments, keeps the state."""

    _tg_name = "?"

    def __new__(cls, *args, **kwargs):
        return object.__new__(cls)

    def __init__(self, *args, **kwargs) -> None:
        pass

    def __setstate__(self, state) -> None:
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and all(isinstance(s, (dict, type(None))) for s in state):
            for part in state:  # (dict state, slots state)
                if part:
                    self.__dict__.update(part)
        else:
            self.__dict__["_state"] = state

    # dict / list subclasses of the source library may be restored item by item
    def __setitem__(self, k, v) -> None:
        self.__dict__.setdefault("_items", {})[k] = v

    def append(self, v) -> None:
        self.__dict__.setdefault("_list", []).append(v)

    def extend(self, vs) -> None:
        self.__dict__.setdefault("_list", []).extend(vs)


def _load_from_bytes_restricted(b: bytes):
    """``torch.storage._load_from_bytes`` is ``torch.load(BytesIO(b), weights_only=False)``: a tensor pickled by the legacy
    (non-zip) protocol would hand its bytes to the UNRESTRICTED unpickler.  Nested payloads re-enter the restricted one."""
    import io

    return torch.load(io.BytesIO(b), map_location="cpu", weights_only=False, pickle_module=_PickleModule)


_PLACEHOLDERS: dict = {}


def _placeholder_for(module: str, name: str):
    key = (module, name)
    if key not in _PLACEHOLDERS:
        _PLACEHOLDERS[key] = type(name, (_Placeholder,), {"_tg_name": f"{module}.{name}"})
    return _PLACEHOLDERS[key]


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if module == "torch_geometric" or module.startswith("torch_geometric."):
            return _placeholder_for(module, name)
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _load_from_bytes_restricted
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or name in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8",
                                                                        "uint8", "bool", "strided", "per_tensor_affine")):
            return getattr(torch, name)
        raise pickle.UnpicklingError(f"graph file names the global {module}.{name}, which a graph file has no business naming "
                                     "(only tensors, arrays, builtin containers and torch_geometric storages are read)")


class _PickleModule:
    """The ``pickle_module`` interface ``torch.load`` wants, with the restricted unpickler."""

    __name__ = "anemoi_core_amd.graphs.io"
    Unpickler = _RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return _RestrictedUnpickler(f, **kwargs).load()

    @staticmethod
    def loads(b, **kwargs):
        import io

        return _RestrictedUnpickler(io.BytesIO(b), **kwargs).load()


class Store(dict):
    """One node or edge store: attribute and item access to the named tensors (the slice of PyG's storage the model reads)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            if k == "num_nodes" and "x" in self:
                return int(self["x"].shape[0])
            raise AttributeError(k) from e

    def __setattr__(self, k, v) -> None:
        self[k] = v


class GraphData:
    """HeteroData-shaped container of plain stores: ``g[name]`` (node store), ``g[(src, relation, dst)]`` (edge store)."""

    def __init__(self, nodes: dict | None = None, edges: dict | None = None, attrs: dict | None = None) -> None:
        self._nodes = {k: Store(v) for k, v in (nodes or {}).items()}
        self._edges = {tuple(k): Store(v) for k, v in (edges or {}).items()}
        self.attrs = dict(attrs or {})

    def __getitem__(self, key):
        table = self._edges if isinstance(key, tuple) else self._nodes
        if key not in table:
            raise KeyError(f"graph has no {'edge' if isinstance(key, tuple) else 'node'} store {key!r}; "
                           f"available: {list(table)}")
        return table[key]

    def __contains__(self, key) -> bool:
        return key in (self._edges if isinstance(key, tuple) else self._nodes)

    def __bool__(self) -> bool:
        return True

    @property
    def node_types(self) -> list:
        return list(self._nodes)

    @property
    def edge_types(self) -> list:
        return list(self._edges)

    def node_items(self) -> list:
        return list(self._nodes.items())

    def edge_items(self) -> list:
        return list(self._edges.items())

    def to(self, device) -> "GraphData":
        mv = lambda s: {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in s.items()}  # noqa: E731
        return GraphData({k: mv(v) for k, v in self._nodes.items()}, {k: mv(v) for k, v in self._edges.items()}, self.attrs)

    def __repr__(self) -> str:
        def fmt(s):
            return ", ".join(f"{k}={tuple(v.shape) if hasattr(v, 'shape') else v!r}" for k, v in s.items())
        lines = [f"  {k}: {fmt(v)}" for k, v in self._nodes.items()] + [f"  {k}: {fmt(v)}" for k, v in self._edges.items()]
        return "GraphData(\n" + "\n".join(lines) + "\n)"


def _mapping_of(store: Any) -> dict:
    """Attribute dict of one storage object in any of the layouts we accept."""
    if isinstance(store, dict):
        return dict(store)
    d = getattr(store, "__dict__", {})
    if isinstance(d.get("_mapping"), dict):  # torch_geometric.data.storage.BaseStorage
        return dict(d["_mapping"])
    if isinstance(d.get("_items"), dict):
        return dict(d["_items"])
    return {k: v for k, v in d.items() if not k.startswith("_")}


def _to_graph_data(obj: Any) -> GraphData:
    if isinstance(obj, GraphData):
        return obj
    if isinstance(obj, dict) and "nodes" in obj and "edges" in obj:  # plain-dict form (what GraphData.save writes)
        return GraphData(obj["nodes"], obj["edges"], obj.get("attrs"))
    d = getattr(obj, "__dict__", None)
    if d is None:
        raise TypeError(f"cannot interpret a {type(obj).__name__} as a graph")
    if "_node_store_dict" in d and "_edge_store_dict" in d:  # torch_geometric.data.HeteroData
        nodes = {k: _mapping_of(v) for k, v in d["_node_store_dict"].items()}
        edges = {tuple(k): _mapping_of(v) for k, v in d["_edge_store_dict"].items()}
        return GraphData(nodes, edges, _mapping_of(d["_global_store"]) if "_global_store" in d else None)
    if "_nodes" in d and "_edges" in d:
        return GraphData({k: _mapping_of(v) for k, v in d["_nodes"].items()}, {tuple(k): _mapping_of(v) for k, v in d["_edges"].items()})
    raise TypeError(f"object of type {getattr(type(obj), '_tg_name', type(obj).__name__)} does not look like a HeteroData "
                    "(no _node_store_dict / _edge_store_dict)")


def load_graph_from_file(graph_filename, map_location="cpu") -> GraphData:
    """Counterpart of the reference's ``load_graph_from_file`` (graphs/src/anemoi/graphs/create.py:179-189) that needs no
    ``torch_geometric``: reads a ``torch.save``d HeteroData (or a plain {"nodes", "edges"} dict) into a ``GraphData``."""
    obj = torch.load(graph_filename, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    return _to_graph_data(obj)


def validate_loaded_graph(graph_data: GraphData, required_dataset_names: list) -> None:
    """graphs/src/anemoi/graphs/create.py:192-200."""
    missing = [n for n in required_dataset_names if n not in graph_data.node_types]
    if missing:
        raise ValueError("Loaded graph is missing dataset node types required by the dataloader. "
                         f"Missing {missing}; available nodes are {graph_data.node_types}.")


def save_graph(graph: GraphData, path) -> None:
    """Plain-dict form of a graph ({"nodes", "edges", "attrs"} of tensors): readable by ``load_graph_from_file`` and by any
    ``torch.load``; the PyG-class pickle itself can only be WRITTEN by torch_geometric."""
    torch.save({"nodes": {k: dict(v) for k, v in graph.node_items()}, "edges": {k: dict(v) for k, v in graph.edge_items()},
                "attrs": graph.attrs}, path)
