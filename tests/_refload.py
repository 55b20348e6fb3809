"""Import helpers for the read-only reference tree (build container only; never on the GPU box).

Nothing here is copied from the reference: the reference's modules are imported from where they lie,
and the un-importable ``train_search.py`` (argparse/mkdir/sys.exit at import) is
AST-sliced so that its pure functions can be executed as the reference wrote them.
"""
import ast
import copy
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'models'))


def import_reference():
    """Returns a namespace with the reference's classes/tables."""
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    ns = types.SimpleNamespace()
    ms = importlib.import_module('models.model_search')
    ns.Network, ns.MixedStage, ns.MixedOP = ms.Network, ms.MixedStage, ms.MixedOP
    ns.model_search = ms
    ns.layers = importlib.import_module('models.layers')
    cfg = importlib.import_module('tools.config')
    ns.mc_mask_dddict, ns.lat_lookup_key_dddict = cfg.mc_mask_dddict, cfg.lat_lookup_key_dddict
    # parsing_model imports model_eval + flops_benchmark (pure torch) -- importable on CPU
    pm = importlib.import_module('parsing_model')
    ns.get_mc_num_dddict, ns.parse_architecture = pm.get_mc_num_dddict, pm.parse_architecture
    ns.get_op_and_depth_weights = pm.get_op_and_depth_weights
    utils = importlib.import_module('tools.utils')
    ns.AverageMeter, ns.accuracy = utils.AverageMeter, utils.accuracy
    return ns


def load_lut(which='gpu'):
    import pickle
    with open(os.path.join(REF, 'latency_pkl', 'latency_%s.pkl' % which), 'rb') as f:
        return pickle.load(f)


def slice_train_search(names, extra_globals=None):
    """exec only the named FunctionDefs of train_search.py in a fresh namespace."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    import logging
    src = open(os.path.join(REF, 'train_search.py')).read()
    ns = {'copy': copy, 'torch': torch, 'nn': nn, 'F': F, 'logging': logging}
    ns.update(extra_globals or {})
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), 'train_search_slice', 'exec'), ns)
    return ns


class inject_gumbel:
    """Context manager: make torch.nn.functional.gumbel_softmax (what models/model_search.py calls through
    its ``F``) consume pre-recorded Exp(1) draws, one row per call, in call order."""

    def __init__(self, rows):
        self.rows = list(rows)
        self.i = 0

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.gumbel_softmax

        def gs(logits, tau=1, hard=False, eps=1e-10, dim=-1):
            e = self.rows[self.i][:logits.numel()].to(logits.dtype)
            self.i += 1
            return ((logits - e.log()) / tau).softmax(dim)
        F.gumbel_softmax = gs
        return self

    def __exit__(self, *a):
        import torch.nn.functional as F
        F.gumbel_softmax = self._orig


def slice_main_epoch_blocks():
    """The three inline blocks of train_search.py's ``main()`` epoch loop, as compiled code objects that run in a caller
    supplied namespace (they are statements, not functions, in the reference):
      'load'    :165-194  current-width model <- max-width state_dict through the masks
      'update'  :234-259  max-width state_dict <- trained model
      'shrink'  :262-307  parse arch, elasticity scaling, L1-norm re-masking  (the body of ``if epoch >= 10``)
    The namespace must provide: model (with .module), state_dict, mc_mask_dddict, mc_maxnum_dddict, lat_lookup,
    lat_lookup_key_dddict, args, logging, torch, np, get_op_and_depth_weights, parse_architecture, get_mc_num_dddict,
    get_lookup_latency, fit_mc_num_by_latency."""
    src = open(os.path.join(REF, 'train_search.py')).read()
    tree = ast.parse(src)
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'main'][0]
    loop = [n for n in main.body if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'epoch'][-1]
    body = loop.body

    def seg(node):
        return ast.get_source_segment(src, node) or ''
    i_load = [i for i, n in enumerate(body) if isinstance(n, ast.For) and "'m_ops' not in key" in seg(n)
              and 'exec(' in seg(n)][0]
    i_upd = [i for i, n in enumerate(body) if isinstance(n, ast.Assign) and 'state_dict_from_model' in seg(n)][0]
    i_shr = [i for i, n in enumerate(body) if isinstance(n, ast.If) and 'epoch >= 10' in seg(n.test)][0]
    blocks = {
        'load': body[i_load:i_load + 2],                       # the two for-loops (non-m_ops keys, then masks)
        'update': body[i_upd:i_upd + 3],                       # state_dict_from_model = ...; two for-loops
        'shrink': body[i_shr].body,
    }
    return {k: compile(ast.fix_missing_locations(ast.Module(v, [])), 'train_search_main_' + k, 'exec')
            for k, v in blocks.items()}


class cuda_is_identity:
    """train_search.py moves index tensors with ``.cuda()``; on the CPU-only build container make that a no-op."""

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *a):
        import torch
        torch.Tensor.cuda = self._orig
