"""Pins the CPU restatement (oracle/awm_oracle.cc) against known-answer vectors captured from the
reference's own test programs (testrandom, testconvcode, scratch harness over the reference's table
builders) -- these were produced with the real libgcrypt AES, not our stand-in."""
import numpy as np
import pytest

import _oracle as orc

ZERO = None


def test_prng_testrandom_vector():
    # testrandom.cc:27-35: zero key, Random(key, 0xf00f1234b00b5678, Stream::bit_order)
    want = [0x8723958e3f2e0422, 0x73187ad3f7275301, 0x17ba2fcebdf91225, 0x2b6c84185eff05a7, 0x8712d5ded1f0886b,
            0x6c02b6e0c05b7af2, 0x74819fa440c7b4e7, 0xdc877bb280d3ee20, 0xc30dbe58838fba92, 0x16d6152aba339588,
            0xd8988a54d29e5ed7, 0x7925937278bfdaa7, 0xcd57c3b2a0648d82, 0xe7826fc24032488a, 0x10ab2e0909d1bc79,
            0x31332c43ff948f27, 0x5d0d35645ee47f3f, 0x83811905e4719f50, 0x56ce264a11a9c9fb, 0x6000867680b8a1b2]
    got = orc.random_u64(ZERO, 0xf00f1234b00b5678, 5, 20)
    assert [int(v) for v in got] == want


def test_prng_sum_of_25m_words():
    # testrandom.cc: after the 20 words and 20 doubles above, the sum of the next 25e6 words mod 2^64
    n = 40 + 25_000_000
    words = orc.random_u64(ZERO, 0xf00f1234b00b5678, 5, n)
    assert int(np.sum(words[40:], dtype=np.uint64)) == 0x8a1c089f52c33da3


def test_random_double_vector():
    want = ["0.973392", "0.367188", "0.714943", "0.331151", "0.209142", "0.913969", "0.278533", "0.373421", "0.998140",
            "0.944676", "0.666369", "0.750097", "0.152569", "0.777862", "0.569157", "0.528917", "0.083540", "0.012448",
            "0.674202", "0.942119"]
    # doubles 21..40 of the same generator: skip 20 words first (one word per double with libstdc++)
    d = orc.random_double(ZERO, 0xf00f1234b00b5678, 5, 40)
    assert ["%f" % v for v in d[20:]] == want


def test_bit_pos_and_up_down_tables():
    pos = orc.bit_pos(ZERO)
    assert pos[:10].tolist() == [160, 1056, 844, 876, 770, 673, 939, 1508, 1810, 69]          # sync_frame(0..9)
    assert pos[510:520].tolist() == [1923, 1500, 519, 1533, 1957, 475, 1315, 1671, 2032, 1644]  # data_frame(0..9)
    up, down = orc.up_down(ZERO, 2, 0)
    assert up.tolist() == [72, 73, 43, 67, 27, 56, 31, 23, 92, 34, 53, 47, 21, 60, 20, 24, 93, 76, 38, 95, 32, 50, 66, 90, 69, 35, 84, 87, 71, 68]
    assert down.tolist() == [63, 39, 30, 57, 99, 94, 41, 28, 36, 22, 55, 78, 77, 79, 45, 33, 46, 89, 98, 80, 48, 44, 74, 54, 61, 64, 100, 25, 26, 62]
    up, down = orc.up_down(ZERO, 2, 509)
    assert up.tolist() == [90, 56, 22, 21, 52, 70, 98, 84, 31, 25, 76, 67, 39, 100, 85, 87, 54, 96, 61, 51, 47, 71, 55, 82, 86, 65, 72, 46, 80, 33]
    up, down = orc.up_down(ZERO, 1, 1)
    assert up.tolist() == [44, 28, 43, 61, 65, 75, 60, 29, 62, 59, 42, 32, 51, 86, 41, 52, 98, 40, 23, 27, 45, 50, 77, 81, 54, 47, 76, 78, 49, 38]
    assert down.tolist() == [64, 56, 33, 34, 31, 83, 80, 96, 88, 67, 72, 57, 85, 82, 87, 94, 92, 99, 79, 26, 68, 58, 69, 36, 22, 70, 53, 55, 46, 84]
    assert orc.bit_pos((42).to_bytes(8, "big") + bytes(8))[:5].tolist() == [485, 1696, 1128, 493, 255]   # --test-key 42


def test_mix_entries():
    e = orc.mix_entries(ZERO)
    assert len(e) == 51480
    assert e[:5].tolist() == [[1143, 95, 66], [1490, 65, 45], [252, 58, 60], [1566, 28, 59], [1371, 82, 44]]
    assert e[-1].tolist() == [1462, 41, 22]


def _hex(bits):
    return "".join("%x" % (bits[i] * 8 + bits[i + 1] * 4 + bits[i + 2] * 2 + bits[i + 3]) for i in range(0, len(bits) - 3, 4))


def test_conv_code_testconvcode_vectors():
    # testconvcode.cc:73-103 with the bits of hex 80f12381
    bits = [int(b) for ch in "80f12381" for b in format(int(ch, 16), "04b")]
    want = {
        0: "fc76c9a37d8902530bbf07dfc06c83b5ebe028c919ed64a9d485669f4dbf76f852f7f0",
        1: "fc77f968916791fbe819140edc155fc9b48c7b1470ea2100306eb5bde61b9c8dbd93f0",
        2: "fff03f3df5c39c4a6ba394974109774f54ca8beb013aa2fef15029b1915fda63ed9ae8501dc5a1921782fce62c218882a72094766d39c7fb74b68bef6b78ead16759eb2fff000",
    }
    for bt, h in want.items():
        coded = orc.conv_encode(bt, bits)
        assert len(coded) == (282 if bt < 2 else 564)
        assert _hex(coded.tolist()) == h
        dec, err = orc.conv_decode_soft(bt, coded.astype(np.float32))
        assert dec.tolist() == bits and err == 0


def test_payload_code_bits():
    payload = [int(b) for ch in "0123456789abcdef0011223344556677" for b in format(int(ch, 16), "04b")]
    order = orc.bit_order(ZERO, 858)
    for bt, first64, shuffled64 in ((0, "00000000003f1dbd", "4e53560c000ae0d4"), (1, "00000000003f1df1", "d942f16f41747402")):
        coded = orc.conv_encode(bt, payload)
        assert len(coded) == 858
        assert _hex(coded[:64].tolist()) == first64
        assert _hex(coded[order][:64].tolist()) == shuffled64            # randomize_bit_order (encode)


def test_sync_bits_and_window():
    sb = orc.sync_bits(ZERO, False)
    assert sb.shape == (6, 85, 61)
    assert sb[0, 0, 0] == 1
    assert sb[0, 0, 1:6].tolist() == [2, 5, 11, 12, 13] and sb[0, 0, 31:36].tolist() == [0, 1, 3, 4, 10]
    w = orc.window(1024)
    assert w[0] == 0 and w[256] == np.float32(0.001953125) and w[512] == np.float32(0.00390625)
    assert abs(float(w[1]) - 3.67670268e-08) < 1e-15 and w[1023] == w[1]
    s = orc.synth_window()
    assert np.count_nonzero(s[:1024]) == 102 and np.count_nonzero(s[2048:]) == 103
    assert s[1024] == 0.5 and np.count_nonzero(s[1024:2048] == 1.0) == 819


def test_noise_generator_matches_reference_wav_samples():
    # test-gen-noise through StdoutWavOutputStream: int(x * 32768) truncated toward zero
    x = orc.gen_noise(ZERO, 8)
    assert [int(v * 32768) for v in x] == [30899, 6421, -26423, 30504, 20179, -12713, -16086, 11788]
