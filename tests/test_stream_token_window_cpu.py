"""s25_stream_kernel fetches a step's 75 token bytes as ALIGNED dwords by LDS-DMA, lane l asking for dword l of
[A - (A & 3), ...).  A fixed 20 dwords per step asked for [end, end + 4) of the buffer on the last game's last step when
A & 3 <= 1 (e.g. B = 4096, K*B*75 % 4 == 0: exactly the dword behind the allocation).  The kernel now requests
nd = ceil(((A & 3) + 75) / 4) dwords; this test restates that count as the source has it and checks, for every alignment
of the step's first token, that the last requested byte is inside the dword that holds the last token -- and that every
token byte the step reads is covered."""
import re
from pathlib import Path

SRC = (Path(__file__).resolve().parent.parent / "mat_mul_amd" / "csrc" / "tg_kernels.hip").read_text()


def kernel_body():
    a = SRC.index("void s25_stream_kernel(StreamArgs a)")
    return SRC[a:SRC.index("\n}\n", a)]


def test_requested_dwords_formula_is_the_one_in_the_kernel():
    body = kernel_body()
    m = re.search(r"const uint32_t nd = \(static_cast<uint32_t>\(A & 3\) \+ (\d+)u \+ (\d+)u\) >> 2;", body)
    assert m, "the per-step dword count of the token DMA is gone: re-derive this test"
    assert (int(m.group(1)), int(m.group(2))) == (75, 3)
    assert "static_cast<uint32_t>(lane) < nd ? 4u * lane : 0u" in body
    assert "lane < 20 ? 4u * lane" not in body          # the fixed 20-dword request is what over-read


def test_no_byte_behind_the_last_tokens_dword_is_requested():
    for base in (0, 4096):                       # a 4-byte aligned buffer
        for B in (1, 2, 3, 4, 4096, 4097):
            for K in (1, 2, 8, 9):
                end = base + 75 * K * B         # one past the last token of the buffer
                for k in range(K):
                    for g in {0, B // 2, B - 1}:
                        A = base + (k * B + g) * 75
                        r = A & 3
                        nd = (r + 75 + 3) >> 2
                        first, last = A - r, A - r + 4 * nd   # [first, last) is requested
                        assert first >= base and first <= A
                        assert last >= A + 75              # every token byte the step reads is in the row
                        assert last <= ((A + 75 + 3) & ~3)  # at most the rest of the last token's dword
                        assert last <= ((end + 3) & ~3)
                        if (end & 3) == 0:
                            assert last <= end             # aligned end: nothing behind the buffer at all
