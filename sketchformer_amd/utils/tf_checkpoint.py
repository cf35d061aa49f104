"""Reading TensorFlow-2 object-based checkpoints (TensorBundle: ``<prefix>.index`` + ``<prefix>.data-0000N-of-0000M``)
WITHOUT TensorFlow, and mapping the reference's variables onto the flat parameter buffers of this package.

The reference saves ``tf.train.Checkpoint(transformer=model, optimizer=optimizer)`` (core/models.py:321-344; the
pretrained models linked at README.md:164 are such checkpoints).  TensorFlow is not available in the build container
and no TF-written file was available either, so this reader is written against the published formats:

* ``.index`` is a LevelDB-format sorted table (tensorflow/core/lib/io/table*.cc): blocks of prefix-compressed
  entries + restart array, each followed by a 5-byte trailer (compression type, masked CRC32C), an index block of
  block handles, a 48-byte footer ending in the magic 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every
  other key a BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto): dtype, shape, shard_id, offset,
  size, crc32c.
* ``.data-*`` shards are the raw little-endian tensor bytes.
* object-graph keys: ``<attribute path from the Checkpoint root>/.ATTRIBUTES/VARIABLE_VALUE``; optimizer slots:
  ``<variable path>/.OPTIMIZER_SLOT/optimizer/<slot>/.ATTRIBUTES/VARIABLE_VALUE``.

``write_tensor_bundle`` produces the same format (used by the tests; it is not needed to train).  Status: format
round-trips and checksums verified against the published CRC32C / varint known answers; NOT yet verified against a
file written by TensorFlow itself.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}
DT_STRING = 7


# ------------------------------------------------------------------ CRC32C (Castagnoli), masked like LevelDB / TF
def _crc_table():
    tab = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return np.array(tab, dtype=np.uint32)


_CRC = _crc_table()


def crc32c(data, crc=0):
    c = (~crc) & 0xffffffff
    tab = _CRC
    for b in memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else data:
        c = int(tab[(c ^ b) & 0xff]) ^ (c >> 8)
    return (~c) & 0xffffffff


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - 0xa282ead8) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------ varints / minimal protobuf
def _put_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _get_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _pb_fields(buf):
    """Yield (field number, wire type, value) of a serialized protobuf message (varint, 64-bit, bytes, 32-bit)."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def parse_bundle_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, wt, v in _pb_fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:                                   # TensorShapeProto
            for n2, _, v2 in _pb_fields(v):
                if n2 == 2:                              # Dim
                    size = 0
                    for n3, _, v3 in _pb_fields(v2):
                        if n3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif num == 7:
            e["sliced"] = True
    return e


def _serialize_entry(dtype_id, shape, shard_id, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# ------------------------------------------------------------------ snappy (blocks may be compressed with it)
def _snappy_uncompress(buf):
    n, pos = _get_varint(buf, 0)
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
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):                             # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ------------------------------------------------------------------ sorted table (LevelDB format)
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    data, ctype, crc = raw[:size], raw[size], struct.unpack("<I", raw[size + 1:size + 5])[0]
    if verify and unmask_crc(crc) != crc32c(raw[:size + 1]):
        raise ValueError("table block at %d: checksum mismatch" % offset)
    if ctype == 1:
        data = _snappy_uncompress(data)
    elif ctype != 0:
        raise ValueError("table block at %d: unknown compression %d" % (offset, ctype))
    return data


def _block_entries(data):
    nrestart = struct.unpack("<I", data[-4:])[0]
    end = len(data) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(data, pos)
        unshared, pos = _get_varint(data, pos)
        vlen, pos = _get_varint(data, pos)
        key = key[:shared] + data[pos:pos + unshared]
        pos += unshared
        yield key, data[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """All (key bytes, value bytes) of a sorted table, in key order."""
    out = []
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        n = f.tell()
        if n < 48:
            raise ValueError("%s: too short for a table" % path)
        f.seek(n - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
            raise ValueError("%s: bad table magic" % path)
        pos = 0
        _, pos = _get_varint(footer, pos)               # metaindex handle
        _, pos = _get_varint(footer, pos)
        ioff, pos = _get_varint(footer, pos)
        isize, pos = _get_varint(footer, pos)
        for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
            boff, p2 = _get_varint(handle, 0)
            bsize, _ = _get_varint(handle, p2)
            out.extend(_block_entries(_read_block(f, boff, bsize, verify)))
    return out


def _build_block(items, restart_interval=16):
    buf, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_table(path, items, block_entries=64):
    """items: sorted list of (key bytes, value bytes)."""
    with open(path, "wb") as f:
        def emit(block):
            off = f.tell()
            f.write(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
            return _put_varint(off) + _put_varint(len(block))
        index = []
        for i in range(0, len(items), block_entries):
            chunk = items[i:i + block_entries]
            index.append((chunk[-1][0], emit(_build_block(chunk))))
        meta = emit(_build_block([]))
        idx = emit(_build_block(index, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ------------------------------------------------------------------ tensor bundle
def read_tensor_bundle(prefix, verify=True, keys=None, verify_tensors=False):
    """{key: ndarray} of every non-string tensor of the checkpoint ``prefix`` (``keys``: optional subset).
    verify: check the CRC32C of every table block; verify_tensors: also of every tensor's bytes (pure-Python CRC: slow
    for the multi-megabyte embedding tables, meant for spot checks)."""
    entries = {}
    num_shards = 1
    for k, v in read_table(prefix + ".index", verify):
        if k == b"":
            for num, _, val in _pb_fields(v):
                if num == 1:
                    num_shards = val
                elif num == 2 and val != 0:
                    raise ValueError("big-endian bundles are not supported")
            continue
        entries[k.decode()] = parse_bundle_entry(v)
    out, shards = {}, {}
    for k, e in entries.items():
        if keys is not None and k not in keys:
            continue
        if e["dtype"] == DT_STRING or e["sliced"]:
            continue                                    # the serialized object graph / partitioned variables: not needed
        if e["dtype"] not in DTYPES:
            raise ValueError("%s: unsupported dtype id %d" % (k, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb")
        f = shards[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if verify_tensors and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: tensor checksum mismatch" % k)
        out[k] = np.frombuffer(raw, dtype=DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    for f in shards.values():
        f.close()
    return out


def list_tensor_bundle(prefix):
    """{key: (dtype id, shape)} without touching the data shards."""
    return {k.decode(): (e["dtype"], tuple(e["shape"]))
            for k, e in ((k, parse_bundle_entry(v)) for k, v in read_table(prefix + ".index") if k != b"")}


def write_tensor_bundle(prefix, tensors):
    """Write {key: ndarray} as a one-shard bundle (test helper / export)."""
    items, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for k in sorted(tensors):
            a = np.asarray(tensors[k])
            a = a if a.flags['C_CONTIGUOUS'] else a.copy(order='C')       # (ascontiguousarray would turn scalars into (1,))
            raw = a.tobytes()
            f.write(raw)
            items.append((k.encode(), _serialize_entry(DTYPE_IDS[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"          # num_shards = 1, version { producer: 1 }
    write_table(prefix + ".index", [(b"", header)] + items)


# ------------------------------------------------------------------ the reference's variable names
def reference_variable_keys(names):
    """Our parameter names (sketchformer_amd / oracle naming) -> object-graph key of the reference's variable, from the
    attribute names of models/sketchformer.py:63-108 and builders/layers/transformer.py (Encoder.enc_layers,
    EncoderLayer.mha / ffn (a Sequential: layer_with_weights-N) / layernormN, SelfAttnV1.W / b / V, DenseExpander.expand_layer)."""
    out = {}
    for n in names:
        p = n.split("/")
        if p[0] in ("encoder", "decoder"):
            base = "transformer/%s" % p[0]
            if p[1] == "embedding":
                key = base + ("/embedding/embeddings" if len(p) == 2 else "/embedding/" + p[2])
            else:
                li = int(p[1][5:])
                base += "/%s_layers/%d" % ("enc" if p[0] == "encoder" else "dec", li)
                if p[2].startswith("mha"):
                    key = "%s/%s/%s/%s" % (base, p[2], p[3], p[4])
                elif p[2] == "ffn":
                    key = "%s/ffn/layer_with_weights-%d/%s" % (base, int(p[3][5:]) - 1, p[4])
                else:
                    key = "%s/%s/%s" % (base, p[2], p[3])
        elif p[0] == "bottleneck":
            if p[1] == "embeding_layer":
                key = "transformer/bottleneck_layer/embeding_layer/" + p[2]
            else:
                key = "transformer/bottleneck_layer/" + p[1].split("_")[0]          # W_attn -> attribute W, ...
        elif p[0] == "class_buffer":
            key = "transformer/class_buffer/%s/%s" % (p[1], p[2])
        elif p[0] == "classify":
            key = "transformer/classify_layer/" + p[1]
        elif p[0] == "expand":
            key = "transformer/expand_layer/expand_layer/" + p[1]
        elif p[0] == "output":
            key = "transformer/output_layer/" + p[1]
        else:
            raise KeyError(n)
        out[n] = key
    return out


def load_reference_checkpoint(prefix, entries, strict=True, verify=True):
    """Read a reference checkpoint and return (params, adam_m, adam_v, scalars): dicts name -> ndarray in our naming
    (``entries`` = engine.param_entries(cfg) or TrainEngine.entries) and {'iterations', 'current_step'} where present.
    Raises with the list of variables that could not be found (strict) - nothing is silently left at its initial value."""
    from ..engine import logical_shape
    names = [e["name"] for e in entries]
    keymap = reference_variable_keys(names)
    have = list_tensor_bundle(prefix)
    want = set()
    for n in names:
        want.add(keymap[n] + SUFFIX)
        for slot in ("m", "v"):
            want.add(keymap[n] + "/.OPTIMIZER_SLOT/optimizer/" + slot + SUFFIX)
    scal_keys = {"iterations": "optimizer/iter" + SUFFIX, "current_step": "transformer/current_step" + SUFFIX}
    data = read_tensor_bundle(prefix, verify=verify, keys=(want | set(scal_keys.values())) & set(have))
    params, m, v, missing = {}, {}, {}, []
    for e in entries:
        n, shape = e["name"], logical_shape(e)
        k = keymap[n] + SUFFIX
        if k not in data:
            missing.append(k)
            continue
        a = data[k]
        if tuple(a.shape) != tuple(shape) and a.size == int(np.prod(shape)):
            a = a.reshape(shape)                       # e.g. V_attn (U,1) / expand kernel (1,L)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("%s: checkpoint shape %r, model shape %r" % (k, a.shape, tuple(shape)))
        params[n] = a.astype(np.float32)
        for slot, dst in (("m", m), ("v", v)):
            ks = keymap[n] + "/.OPTIMIZER_SLOT/optimizer/" + slot + SUFFIX
            if ks in data:
                dst[n] = data[ks].reshape(shape).astype(np.float32)
    if missing and strict:
        raise KeyError("variables not found in %s (%d of %d): %s ...; keys in the file look like: %s"
                       % (prefix, len(missing), len(names), missing[:4], sorted(have)[:4]))
    scalars = {k: int(data[key]) for k, key in scal_keys.items() if key in data}
    return params, m, v, scalars
