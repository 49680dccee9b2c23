"""Philox4x32-10: both copies (oracle and engine) against the known-answer
vectors published with the algorithm (Salmon et al., SC'11; Random123
kat_vectors), and the shared draw convention."""
from oracle import oracle_ffi as O

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_oracle_philox_known_answers():
    for ctr, key, want in KAT:
        assert tuple(O.philox(ctr, key)) == want


def test_draws_are_in_unit_interval_and_keyed_by_id():
    seen = set()
    for hid in (0, 1, 2**32 + 5):
        for k in range(6):
            u = O.lib().r3d_oracle_draw(0x5EED, hid, k)
            assert 0.0 < u <= 1.0
            seen.add(u)
    assert len(seen) == 18
    # draw k uses words [2(k&1), 2(k&1)+1] of block k>>1 with counter (id_lo, id_hi, k>>1, 0)
    w = O.philox((7, 0, 1, 0), (0x5EED, 0))
    m = ((w[2] >> 5) << 26) | (w[3] >> 6)
    assert O.lib().r3d_oracle_draw(0x5EED, 7, 3) == (m + 1) / 2.0**53
