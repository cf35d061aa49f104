"""TensorFlow checkpoint (TensorBundle) reader: format pieces against published known answers, write/read round trips,
and the mapping of the reference's object-graph keys onto this package's parameter names.
No TF-written file is available in the build container: that last link stays unverified and the module says so."""
import struct

import numpy as np
import pytest

from sketchformer_amd.utils import tf_checkpoint as tfc


def test_crc32c_known_answers_and_masking():
    assert tfc.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value (RFC 3720 appendix B.4)
    assert tfc.crc32c(b"\x00" * 32) == 0x8A9136AA                     # RFC 3720 B.4: 32 bytes of zeros
    assert tfc.crc32c(b"\xff" * 32) == 0x62A8AB43                     # 32 bytes of 0xff
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E                 # incrementing bytes
    c = tfc.crc32c(b"foo")
    assert tfc.unmask_crc(tfc.mask_crc(c)) == c and tfc.mask_crc(c) != c
    assert tfc.crc32c(b"hello world") == tfc.crc32c(b" world", tfc.crc32c(b"hello"))   # incremental


def test_varint_and_entry_proto_round_trip():
    for n in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 62):
        v, pos = tfc._get_varint(tfc._put_varint(n), 0)
        assert v == n and pos == len(tfc._put_varint(n))
    assert tfc._put_varint(300) == b"\xac\x02"                        # protobuf encoding guide example
    raw = tfc._serialize_entry(1, (1004, 128), 0, 4096, 1004 * 128 * 4, 0xdeadbeef)
    e = tfc.parse_bundle_entry(raw)
    assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"]) == (1, [1004, 128], 4096, 1004 * 128 * 4, 0xdeadbeef)
    assert tfc.parse_bundle_entry(tfc._serialize_entry(9, (), 0, 0, 8, 1))["shape"] == []


def test_snappy_decoder():
    # literal "abcd" + copy(offset 4, length 8) -> "abcdabcdabcd"; preamble = uncompressed length 12
    stream = bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1 | (0 << 5), 4])
    assert tfc._snappy_uncompress(stream) == b"abcdabcdabcd"
    # 2-byte-offset copy with an overlapping run: "ab" then copy(offset 2, length 6)
    stream = bytes([8, (2 - 1) << 2]) + b"ab" + bytes([((6 - 1) << 2) | 2]) + struct.pack("<H", 2)
    assert tfc._snappy_uncompress(stream) == b"abababab"


def test_table_and_bundle_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {}
    for i in range(300):                                               # several data blocks, long shared prefixes
        tensors["transformer/encoder/enc_layers/%d/mha/wq/kernel%s" % (i, tfc.SUFFIX)] = rng.randn(3, 5).astype(np.float32)
    tensors["optimizer/iter" + tfc.SUFFIX] = np.array(12345, dtype=np.int64)
    tensors["a/bool"] = np.array([True, False])
    tensors["z/f64"] = rng.randn(7)
    prefix = str(tmp_path / "ckpt-1")
    tfc.write_tensor_bundle(prefix, tensors)
    keys = [k for k, _ in tfc.read_table(prefix + ".index")]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) == len(tensors) + 1
    got = tfc.read_tensor_bundle(prefix, verify_tensors=True)
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape and np.array_equal(got[k], tensors[k])
    assert tfc.list_tensor_bundle(prefix)["optimizer/iter" + tfc.SUFFIX] == (9, ())
    # corruption is detected: flip one byte of the index / of a tensor
    blob = bytearray(open(prefix + ".index", "rb").read())
    blob[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        tfc.read_table(prefix + ".index")
    tfc.write_tensor_bundle(prefix, tensors)
    blob = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    blob[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        tfc.read_tensor_bundle(prefix, verify_tensors=True)


@pytest.mark.parametrize("kw", [dict(), dict(continuous=True), dict(attn_version=2, class_buffer_layers=1)])
def test_reference_checkpoint_maps_onto_our_parameters(tmp_path, kw):
    """A bundle laid out like the reference's tf.train.Checkpoint(transformer=..., optimizer=...) loads into our names:
    every variable, both Adam slots, optimizer.iterations and current_step."""
    import oracle
    from sketchformer_amd import engine
    base = dict(num_layers=2, d_model=64, dff=128, num_heads=4, lowerdim=64, vocab_size=52, n_classes=7, seq_len=24)
    base.update(kw)
    ocfg = oracle.Config(**base)
    mk = dict(base)
    if mk.get("continuous"):
        mk["vocab_size"] = None
    entries = engine.param_entries(engine.make_config(batch=4, **mk))
    P = oracle.init_params(ocfg, seed=3, dtype=np.float32)
    keymap = tfc.reference_variable_keys(list(P))
    assert keymap["encoder/layer1/ffn/dense2/bias"] == "transformer/encoder/enc_layers/1/ffn/layer_with_weights-1/bias"
    assert keymap["decoder/layer0/mha2/wk/kernel"] == "transformer/decoder/dec_layers/0/mha2/wk/kernel"
    assert keymap["bottleneck/V_attn"] == "transformer/bottleneck_layer/V"
    assert keymap["expand/kernel"] == "transformer/expand_layer/expand_layer/kernel"
    rng = np.random.RandomState(1)
    tensors, M, V = {}, {}, {}
    for n, a in P.items():
        tensors[keymap[n] + tfc.SUFFIX] = a
        M[n], V[n] = rng.randn(*a.shape).astype(np.float32), rng.rand(*a.shape).astype(np.float32)
        tensors[keymap[n] + "/.OPTIMIZER_SLOT/optimizer/m" + tfc.SUFFIX] = M[n]
        tensors[keymap[n] + "/.OPTIMIZER_SLOT/optimizer/v" + tfc.SUFFIX] = V[n]
    tensors["optimizer/iter" + tfc.SUFFIX] = np.array(4321, dtype=np.int64)
    tensors["transformer/current_step" + tfc.SUFFIX] = np.array(4320, dtype=np.int64)
    tensors["optimizer/beta_1" + tfc.SUFFIX] = np.array(0.9, dtype=np.float32)          # extra keys are ignored
    prefix = str(tmp_path / "ckpt-7")
    tfc.write_tensor_bundle(prefix, tensors)
    params, m, v, scalars = tfc.load_reference_checkpoint(prefix, entries)
    assert set(params) == set(P) == set(m) == set(v)
    for n in P:
        assert np.array_equal(params[n], P[n]) and np.array_equal(m[n], M[n]) and np.array_equal(v[n], V[n])
    assert scalars == {"iterations": 4321, "current_step": 4320}
    del tensors[keymap["output/bias"] + tfc.SUFFIX]
    tfc.write_tensor_bundle(prefix, tensors)
    with pytest.raises(KeyError, match="output_layer/bias"):
        tfc.load_reference_checkpoint(prefix, entries)
