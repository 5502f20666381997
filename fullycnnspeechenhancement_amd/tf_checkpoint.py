"""TensorFlow-free readers for the files the reference stores weights in.

The reference fills its session with `tf.train.Saver(...).restore(sess, checkpoint_file)`
(model_utils/tester.py:36-39, trainer.py:51,61) and exports a frozen GraphDef with
`convert_variables_to_constants` (freeze.py:31-48).  TensorFlow 1.14 is not installable here, so this
module restates the two on-disk formats from their published definitions and reads them with numpy only:

* V2 checkpoint ("tensor bundle", the Saver default since TF 1.0): `<prefix>.index` is a
  LevelDB-format sorted string table (tensorflow/core/lib/io/table*.cc, format.cc) whose values are
  `BundleHeaderProto` (key "") and `BundleEntryProto` (key = variable name)
  (tensorflow/core/protobuf/tensor_bundle.proto); tensors are raw little-endian bytes in
  `<prefix>.data-SSSSS-of-NNNNN` at the entry's (shard, offset, size), protected by a masked CRC-32C.
* Frozen graph `.pb`: a serialized `GraphDef`; variables become `Const` nodes that keep the variable's
  name, with the value in attr "value" -> `TensorProto` (tensor_content or float_val).

Parity note: no TF-written file exists in /root/reference or this image, so the readers are pinned only to
the format definitions and to the writers below (round trip) -- "unpinned" against real TF output.
The writers emit files TF itself can restore (uncompressed blocks, one shard), which is how a user can go
back from this framework's trained variables to the reference.
"""

import os
import struct

import numpy as np

from . import spec

# ---------------------------------------------------------------------------------------------
# protobuf wire format (only what the three messages need)
# ---------------------------------------------------------------------------------------------


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _write_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """Yield (field_number, wire_type, value) over a serialized message; value is int or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            if len(v) != ln:
                raise ValueError("truncated length-delimited field")
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _field(num, wt, payload):
    key = _write_varint((num << 3) | wt)
    if wt == 0:
        return key + _write_varint(payload)
    if wt == 2:
        return key + _write_varint(len(payload)) + payload
    return key + payload


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_ENUM = {np.dtype(v): k for k, v in _DTYPES.items()}


def _parse_shape(buf):
    dims = []
    for num, _, v in _fields(buf):
        if num == 2:                       # repeated Dim dim
            size = 0
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return tuple(dims)


def _shape_proto(shape):
    return b"".join(_field(2, 2, _field(1, 0, int(d))) for d in shape)


# ---------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli), masked as LevelDB / TF do
# ---------------------------------------------------------------------------------------------

_CRC_TABLE = None


def _crc_tables():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t0 = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t0[i] = c
        tabs = [t0]
        for _ in range(7):                 # slicing-by-8 tables
            prev = tabs[-1]
            tabs.append((prev >> 8) ^ t0[prev & 0xFF])
        _CRC_TABLE = [t.tolist() for t in tabs]
    return _CRC_TABLE


def crc32c(data, crc=0):
    t = _crc_tables()
    t0, t1, t2, t3, t4, t5, t6, t7 = t
    c = crc ^ 0xFFFFFFFF
    mv = memoryview(data).cast("B")
    n8 = len(mv) // 8 * 8
    for i in range(0, n8, 8):
        lo = c ^ (mv[i] | mv[i + 1] << 8 | mv[i + 2] << 16 | mv[i + 3] << 24)
        c = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24] ^
             t3[mv[i + 4]] ^ t2[mv[i + 5]] ^ t1[mv[i + 6]] ^ t0[mv[i + 7]])
    for i in range(n8, len(mv)):
        c = t0[(c ^ mv[i]) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------
# snappy block decompression (index blocks may be snappy-compressed by some writers)
# ---------------------------------------------------------------------------------------------


def _snappy_decompress(buf):
    n, pos = _read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | buf[pos + 1] << 8
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# LevelDB-format table (the .index file)
# ---------------------------------------------------------------------------------------------

_TABLE_MAGIC = 0xDB4775248B80FB57
_FOOTER_LEN = 48


def _read_block(data, offset, size, verify):
    contents = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(contents) != size or len(trailer) != 5:
        raise ValueError("table block out of range")
    if verify:
        want = struct.unpack("<I", trailer[1:])[0]
        if _mask_crc(crc32c(contents + trailer[:1])) != want:
            raise ValueError("table block checksum mismatch")
    if trailer[0] == 1:
        contents = _snappy_decompress(contents)
    elif trailer[0] != 0:
        raise ValueError("unknown block compression %d" % trailer[0])
    return contents


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("table block too small")
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < _FOOTER_LEN or struct.unpack("<Q", data[-8:])[0] != _TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-_FOOTER_LEN:]
    pos = 0
    _, pos = _read_varint(footer, pos)          # metaindex handle (unused)
    _, pos = _read_varint(footer, pos)
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


def _build_block(entries, restart_interval=16):
    buf, restarts, last, count = bytearray(), [], b"", 0
    for key, value in entries:
        if count % restart_interval == 0:
            restarts.append(len(buf))
            shared = 0
        else:
            shared = 0
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        buf += _write_varint(shared) + _write_varint(len(key) - shared) + _write_varint(len(value))
        buf += key[shared:] + value
        last, count = key, count + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_table(path, items, block_size=4096):
    """Write sorted (key, value) byte pairs as an uncompressed LevelDB-format table."""
    items = sorted(items)
    out = bytearray()

    def emit(block):
        handle = _write_varint(len(out)) + _write_varint(len(block))
        trailer = b"\x00"
        out.extend(block + trailer + struct.pack("<I", _mask_crc(crc32c(block + trailer))))
        return handle

    index, pending, pending_size = [], [], 0
    for key, value in items:
        pending.append((key, value))
        pending_size += len(key) + len(value) + 8
        if pending_size >= block_size:
            index.append((pending[-1][0], emit(_build_block(pending))))
            pending, pending_size = [], 0
    if pending:
        index.append((pending[-1][0], emit(_build_block(pending))))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as fh:
        fh.write(bytes(out))


# ---------------------------------------------------------------------------------------------
# tensor bundle = V2 checkpoint
# ---------------------------------------------------------------------------------------------


def _shard_name(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def checkpoint_prefix(path):
    """Accept what users pass to Saver.restore (the prefix) or any of the bundle's files."""
    for suffix in (".index", ".meta"):
        if path.endswith(suffix):
            return path[:-len(suffix)]
    head, sep, tail = path.rpartition(".data-")
    if sep and "-of-" in tail:
        return head
    return path


def read_checkpoint(path, verify=True, names=None):
    """{variable name: ndarray} from a TF V2 checkpoint.  `names`: optional iterable to restrict to."""
    prefix = checkpoint_prefix(path)
    index = prefix + ".index"
    if not os.path.exists(index):
        raise FileNotFoundError("no checkpoint index at %s" % index)
    entries = read_table(index, verify)
    num_shards, shards, out = 1, {}, {}
    wanted = None if names is None else set(names)
    for key, value in entries:
        if key == b"":
            for num, _, v in _fields(value):
                if num == 1:
                    num_shards = v
                elif num == 2 and v != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        name = key.decode("utf-8")
        if wanted is not None and name not in wanted:
            continue
        dtype, shape, shard, offset, size, crc, sliced = 0, (), 0, 0, 0, None, False
        for num, wt, v in _fields(value):
            if num == 1:
                dtype = v
            elif num == 2:
                shape = _parse_shape(v)
            elif num == 3:
                shard = v
            elif num == 4:
                offset = _signed64(v)
            elif num == 5:
                size = _signed64(v)
            elif num == 6:
                crc = struct.unpack("<I", v)[0]
            elif num == 7:
                sliced = True
        if sliced:
            raise ValueError("variable %r is stored as slices (partitioned variable): not supported" % name)
        if dtype not in _DTYPES:
            raise ValueError("variable %r has unsupported dtype enum %d" % (name, dtype))
        if shard not in shards:
            with open(_shard_name(prefix, shard, num_shards), "rb") as fh:
                shards[shard] = fh.read()
        raw = shards[shard][offset:offset + size]
        np_dtype = np.dtype(_DTYPES[dtype])
        count = int(np.prod(shape)) if shape else 1
        if len(raw) != size or size != count * np_dtype.itemsize:
            raise ValueError("variable %r: %d bytes on disk, shape %s needs %d" % (name, len(raw), shape, count * np_dtype.itemsize))
        if verify and crc is not None and _mask_crc(crc32c(raw)) != crc:
            raise ValueError("variable %r: data checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=np_dtype.newbyteorder("<")).astype(np_dtype).reshape(shape)
    return out


def write_checkpoint(prefix, tensors):
    """Write {name: ndarray} as a one-shard V2 checkpoint (prefix.index + prefix.data-00000-of-00001)."""
    data = bytearray()
    items = [(b"", _field(1, 0, 1) + _field(2, 0, 0) + _field(3, 2, _field(1, 0, 1)))]   # 1 shard, little endian, producer 1
    for name in sorted(tensors):
        a = np.asarray(tensors[name])          # (ascontiguousarray would turn a scalar into shape (1,))
        if a.dtype not in _DTYPE_ENUM:
            raise ValueError("variable %r: dtype %s cannot be stored" % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes(order="C")
        entry = _field(1, 0, _DTYPE_ENUM[a.dtype]) + _field(2, 2, _shape_proto(a.shape))
        if len(data):
            entry += _field(4, 0, len(data))
        entry += _field(5, 0, len(raw)) + _field(6, 5, struct.pack("<I", _mask_crc(crc32c(raw))))
        items.append((name.encode("utf-8"), entry))
        data += raw
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(_shard_name(prefix, 0, 1), "wb") as fh:
        fh.write(bytes(data))
    write_table(prefix + ".index", items)


# ---------------------------------------------------------------------------------------------
# frozen GraphDef (.pb)
# ---------------------------------------------------------------------------------------------


def _parse_tensor_proto(buf):
    dtype, shape, content, floats, ints = 0, (), None, [], []
    for num, wt, v in _fields(buf):
        if num == 1:
            dtype = v
        elif num == 2:
            shape = _parse_shape(v)
        elif num == 4:
            content = v
        elif num == 5:                                  # float_val, packed or not
            floats.extend(struct.unpack("<%df" % (len(v) // 4), v) if wt == 2 else struct.unpack("<f", v))
        elif num == 7:                                  # int_val
            if wt == 2:
                pos = 0
                while pos < len(v):
                    x, pos = _read_varint(v, pos)
                    ints.append(_signed64(x))
            else:
                ints.append(_signed64(v))
    if dtype not in _DTYPES:
        return None
    np_dtype = np.dtype(_DTYPES[dtype])
    count = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        return np.frombuffer(content, dtype=np_dtype.newbyteorder("<")).astype(np_dtype).reshape(shape)
    vals = floats if floats else ints
    if not vals:
        return np.zeros(shape, np_dtype)
    a = np.asarray(vals, dtype=np_dtype)
    if a.size == 1 and count != 1:                      # TF stores a splat as one value
        a = np.full(count, a[0], np_dtype)
    return a.reshape(shape)


def read_frozen_graph(path):
    """{node name: ndarray} for every Const node of a serialized GraphDef (freeze.py:43-48 output)."""
    with open(path, "rb") as fh:
        data = fh.read()
    out = {}
    for num, wt, node in _fields(data):
        if num != 1 or wt != 2:
            continue
        name, op, value = "", "", None
        for n2, _, v2 in _fields(node):
            if n2 == 1:
                name = v2.decode("utf-8")
            elif n2 == 2:
                op = v2.decode("utf-8")
            elif n2 == 5:                               # map<string, AttrValue> entry
                k, av = None, None
                for n3, _, v3 in _fields(v2):
                    if n3 == 1:
                        k = v3
                    elif n3 == 2:
                        av = v3
                if k == b"value" and av is not None:
                    for n4, _, v4 in _fields(av):
                        if n4 == 8:                     # AttrValue.tensor
                            value = v4
        if op == "Const" and value is not None:
            t = _parse_tensor_proto(value)
            if t is not None:
                out[name] = t
    return out


def write_frozen_graph(path, tensors):
    """Minimal GraphDef of Const nodes (what freezing leaves of the variables); enough for TF to import."""
    out = bytearray()
    for name in sorted(tensors):
        a = np.asarray(tensors[name])          # (ascontiguousarray would turn a scalar into shape (1,))
        tp = (_field(1, 0, _DTYPE_ENUM[a.dtype]) + _field(2, 2, _shape_proto(a.shape)) +
              _field(4, 2, a.astype(a.dtype.newbyteorder("<")).tobytes(order="C")))
        attr_value = _field(5, 2, _field(1, 2, b"value") + _field(2, 2, _field(8, 2, tp)))
        attr_dtype = _field(5, 2, _field(1, 2, b"dtype") + _field(2, 2, _field(6, 0, _DTYPE_ENUM[a.dtype])))
        node = _field(1, 2, name.encode("utf-8")) + _field(2, 2, b"Const") + attr_dtype + attr_value
        out += _field(1, 2, node)
    with open(path, "wb") as fh:
        fh.write(bytes(out))


# ---------------------------------------------------------------------------------------------
# the reference's variables
# ---------------------------------------------------------------------------------------------

ADAM_SLOTS = ("Adam", "Adam_1")          # tf.train.AdamOptimizer slot names: m, v  (trainer.py:177)


def load_reference_weights(path, variant):
    """Weights dict for `variant` from a .npz, a TF V2 checkpoint (prefix or .index) or a frozen .pb.

    Returns only the model variables (module.py:27,29 names); optimizer slots and counters that
    `Saver(tf.global_variables())` also stores are ignored here -- see `load_training_state`.
    """
    names = [n for n, _ in spec.variable_shapes(variant)]
    if path.endswith(".npz"):
        with np.load(path) as z:
            found = {k: z[k] for k in z.files}
    elif path.endswith(".pb"):
        found = read_frozen_graph(path)
    else:
        found = read_checkpoint(path, names=names)
    missing = [n for n in names if n not in found]
    if missing:
        raise KeyError("%s lacks %d of %d variables of this network, e.g. %r (wrong net_work?)" %
                       (path, len(missing), len(names), missing[0]))
    return {n: np.asarray(found[n], dtype=np.float32) for n in names}


def load_training_state(path, variant):
    """(weights, adam_m, adam_v, global_step) from a checkpoint written by the reference trainer
    (trainer.py:51 saves every global variable: model, `<var>/Adam`, `<var>/Adam_1`, `global_step`).
    Slots are None when absent (e.g. an inference-only export)."""
    allv = read_checkpoint(path)
    names = [n for n, _ in spec.variable_shapes(variant)]
    weights = {n: np.asarray(allv[n], np.float32) for n in names}
    slots = []
    for slot in ADAM_SLOTS:
        keys = ["%s/%s" % (n, slot) for n in names if not n.endswith(("moving_mean", "moving_variance"))]
        slots.append({k.rsplit("/", 1)[0]: np.asarray(allv[k], np.float32) for k in keys} if all(k in allv for k in keys) else None)
    step = int(allv["global_step"]) if "global_step" in allv else None
    # tf.train.AdamOptimizer keeps beta1^(t+1) after t applied updates in `beta1_power`; the library derives Adam's t
    # from the step counter, so the two must agree (they do for every checkpoint the reference trainer writes:
    # train_op bumps global_step once per update, trainer.py:178).  A checkpoint without global_step resumes at
    # the t that beta1_power implies.
    # beta1_power is a float32: 0.9^(t+1) leaves the normal range near t = 830 and underflows to 0 near t = 980, so it is
    # trusted only while it is comfortably normal; past that beta2_power gives the count (it falls below the 1e-30 cut-off
    # near t = 69 k), and a checkpoint from which it cannot be recovered says so instead of silently resuming at step 0.
    # TF forms both powers by REPEATED float32 multiplication with float32(0.9) = 0.89999998 / float32(0.999) = 0.99900001,
    # so the logarithm is taken to THOSE bases (with log(0.999) the estimate drifts by 1.3e-5 per step: one step off at
    # t = 40 k), and what remains of the float32 rounding of ~t products is allowed for: |t_adam - global_step| <=
    # 1 + 1e-4 * step passes without a warning.
    t_adam, used = None, None
    for key, base in (("beta1_power", np.float32(0.9)), ("beta2_power", np.float32(0.999))):
        if t_adam is None and key in allv:
            bp = float(np.asarray(allv[key]).reshape(-1)[0])
            if 1e-30 < bp < 1.0:
                t_adam, used = int(round(np.log(bp) / np.log(float(base)))) - 1, key
    if t_adam is None and ("beta1_power" in allv or "beta2_power" in allv) and "global_step" not in allv:
        import warnings
        warnings.warn("%s: the Adam update count cannot be recovered (beta1_power / beta2_power have underflowed and there is "
                      "no global_step); resuming at step 0 resets Adam's bias correction and the Noam learning rate" % path)
    if step is None:
        step = max(t_adam, 0) if t_adam is not None else 0
    elif t_adam is not None and abs(t_adam - step) > 1 + 1e-4 * step:
        import warnings
        warnings.warn("%s: %s implies %d Adam updates but global_step is %d; resuming at global_step"
                      % (path, used, t_adam, step))
    return weights, slots[0], slots[1], step


def write_checkpoint_state(prefix):
    """The `checkpoint` state file tf.train.latest_checkpoint reads (CheckpointState text proto), next to the
    bundle: the reference's `continue_train` finds its checkpoint through it (trainer.py:52-58)."""
    d = os.path.dirname(os.path.abspath(prefix))
    name = os.path.basename(prefix)
    with open(os.path.join(d, "checkpoint"), "w") as fh:
        fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
