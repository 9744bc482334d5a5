"""Checkpoint file of the reference (tools/utils.py:567-586 save_model, :323-345 resume):

    {'state_dict', 'epoch', 'iter', 'buffer', 'buffer_cnt', 'loss_data'}

`buffer`/`buffer_cnt` are the intertwiner's history of big-object class features as numpy
arrays ([] when the intertwiner is off).  Module names and parameter shapes are the
reference's (tests/golden/state_dict_keys.json pins them), so the authors' `.pth` files load
by name; like the reference, loading is non-strict and also accepts a bare state dict (its
"legacy / pretrain" case).  The Keras weight importer is feature_intertwiner_amd/tools/convert_from_keras.py.
"""
import numpy as np
import torch

from .intertwiner import FeatureBuffer


def save_model(model, path, epoch, iter, loss_data=None):
    cfg = model.config
    if cfg.DEV.SWITCH and not cfg.DEV.BASELINE and model.feature_buffer is not None:
        buffer = model.feature_buffer.buffer.cpu().numpy()
        buffer_cnt = model.feature_buffer.buffer_cnt.cpu().numpy()
    else:
        buffer, buffer_cnt = [], []
    torch.save({'state_dict': model.state_dict(), 'epoch': int(epoch), 'iter': int(iter),
                'buffer': buffer, 'buffer_cnt': buffer_cnt,
                'loss_data': [] if loss_data is None else loss_data}, path)


def _numpy_safe_globals():
    """Allow-list for the two ndarrays in the file.  torch matches a pickled global by the STRING
    `module.name` stored in the file, and the reference's checkpoints were written under numpy 1.x, where
    the array constructor pickles as 'numpy.core.multiarray._reconstruct'; numpy 2 renamed the module to
    'numpy._core.multiarray' (the old path is a deprecated alias whose functions report the new
    __module__), so both spellings are registered explicitly as (callable, 'full.path') pairs."""
    try:
        import numpy._core.multiarray as ma          # numpy >= 2
    except ImportError:                               # numpy 1.x
        import numpy.core.multiarray as ma
    safe = [np.ndarray, np.dtype,
            (ma._reconstruct, "numpy.core.multiarray._reconstruct"),
            (ma._reconstruct, "numpy._core.multiarray._reconstruct")]
    if hasattr(ma, "scalar"):                         # 0-d numpy scalars (e.g. an np.float64 in loss_data)
        safe += [(ma.scalar, "numpy.core.multiarray.scalar"), (ma.scalar, "numpy._core.multiarray.scalar")]
    # numpy >= 1.25 pickles dtypes through their per-type classes (numpy.dtypes.Float32DType ...)
    safe += list({type(np.dtype(t)) for t in (np.float16, np.float32, np.float64, np.int8, np.uint8, np.int16,
                                               np.int32, np.int64, np.bool_)})
    return safe


def _read(path, map_location, trusted):
    """The file holds tensors, python scalars/lists and two numpy arrays.  It is read with the
    restricted unpickler (weights_only=True, numpy's array reconstruction allow-listed); a
    third-party .pth that needs arbitrary pickles loads only with trusted=True."""
    if trusted:
        return torch.load(path, map_location=map_location, weights_only=False)
    safe = _numpy_safe_globals()
    with torch.serialization.safe_globals(safe):
        return torch.load(path, map_location=map_location, weights_only=True)


def load_model(model, path, map_location=None, trusted=False):
    """Returns (start_epoch, start_iter, loss_data, missing_keys, unexpected_keys).  A resumed
    model continues at iter+1 (epoch roll-over needs the dataset size and is left to the caller,
    tools/utils.py:333-339); a bare state dict is a pretrain model and starts at (1, 1).  The
    history buffer is adopted only if its length equals cfg.DEV.BUFFER_SIZE; otherwise it is
    re-initialised, as tools/utils.py:379-384 does."""
    from .data_parallel import invalidate_derived_state
    ckpt = _read(path, map_location, trusted)
    state = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    result = model.load_state_dict(state, strict=False)
    invalidate_derived_state(model)
    if not (isinstance(ckpt, dict) and 'epoch' in ckpt and 'iter' in ckpt):
        return 1, 1, [], result.missing_keys, result.unexpected_keys
    buf = ckpt.get('buffer', [])
    if isinstance(buf, np.ndarray) and buf.size:
        dev = next(model.parameters()).device
        if buf.shape[0] == int(model.config.DEV.BUFFER_SIZE):
            fb = FeatureBuffer(buf.shape[0], buf.shape[1], buf.shape[2], dev)
            fb.buffer = torch.from_numpy(buf).to(dev)
            fb.buffer_cnt = torch.from_numpy(np.asarray(ckpt['buffer_cnt'])).to(dev)
            model.feature_buffer = fb
        else:
            model.initialize_buffer(dev)
    return int(ckpt['epoch']), int(ckpt['iter']) + 1, ckpt.get('loss_data', []), result.missing_keys, result.unexpected_keys
