"""not gpu: the algebra the TILT kernel class relies on, checked in plain Python integers.

With g = the interior gap extension (both sides), X*(i,j) = X(i,j) + (i+j) g turns the search16 recurrence
into the same recurrence with score' = score + 2g, QR' = QR - g, R' = R - g and shifted borders.  The test runs both forms
on random sequences / penalties and requires, for every cell: H* - (i+j) g == H, identical direction bits, and the bounds
the compressed checkpoints assume (H - F_next and H - E_next inside [min(R', QR'), QR'])."""
import random

import pytest


def _score(a, b, match, mismatch):
    return match if a == b else mismatch


def _dp(q, t, match, mismatch, pen, tilt):
    """pen = dict(go/ge for q|t x l|i|r).  Returns (H, bits, dF, dE) with H un-tilted; arithmetic without saturation."""
    Q, D = len(q), len(t)
    g = pen["ge_t_i"] if tilt else 0
    QR = lambda x, y: pen[f"go_{x}_{y}"] + pen[f"ge_{x}_{y}"] - g
    R = lambda x, y: pen[f"ge_{x}_{y}"] - g
    # borders in (possibly tilted) coordinates: X*(i,j) = X(i,j) + (i+j) g
    htop = [-(pen["go_q_l"] + (j + 1) * pen["ge_q_l"]) + (j - 1) * g for j in range(D)]          # row -1
    hleft = [-(pen["go_t_l"] + (i + 1) * pen["ge_t_l"]) + (i - 1) * g for i in range(Q)]         # column -1
    corner = 0 - 2 * g
    H = [[0] * D for _ in range(Q)]
    bits = [[0] * D for _ in range(Q)]
    dF = [[0] * D for _ in range(Q)]
    dE = [[0] * D for _ in range(Q)]
    E = [hleft[i] - QR("q", "i" if i < Q - 1 else "r") for i in range(Q)]                        # E(i, 0)
    for j in range(D):
        qrt, rt = (QR("t", "i"), R("t", "i")) if j < D - 1 else (QR("t", "r"), R("t", "r"))
        F = htop[j] - qrt                                                                          # F(0, j)
        for i in range(Q):
            qrq, rq = (QR("q", "i"), R("q", "i")) if i < Q - 1 else (QR("q", "r"), R("q", "r"))
            if i == 0:
                hd = corner if j == 0 else htop[j - 1]
            else:
                hd = hleft[i - 1] if j == 0 else H[i - 1][j - 1]
            h = hd + _score(q[i], t[j], match, mismatch) + 2 * g
            up = F > h
            h = max(h, F)
            left = E[i] > h
            h = max(h, E[i])
            hf, f = h - qrt, F - rt
            he, e = h - qrq, E[i] - rq
            bits[i][j] = (up, left, f > hf, e > he)
            F = max(f, hf)
            E[i] = max(e, he)
            H[i][j] = h
            dF[i][j], dE[i][j] = h - F, h - E[i]
    if tilt:
        # the stored values are tilted; hand back plain H for the comparison
        Hp = [[H[i][j] - (i + j) * g for j in range(D)] for i in range(Q)]
        return Hp, bits, dF, dE
    return H, bits, dF, dE


@pytest.mark.parametrize("seed", range(6))
def test_tilted_recurrence_is_the_same_alignment(seed, oracle):
    rng = random.Random(seed)
    for _ in range(40):
        g = rng.choice([1, 2, 3, 5])
        pen = {"ge_q_i": g, "ge_t_i": g}
        go_i = rng.choice([0, 3, 10, 18, 30])
        pen["go_q_i"] = pen["go_t_i"] = go_i
        for side in "qt":
            for end in "lr":
                pen[f"go_{side}_{end}"] = rng.choice([0, 1, 2, 20])
                pen[f"ge_{side}_{end}"] = rng.choice([0, 1, 2, 4])
        match, mismatch = rng.choice([(2, -4), (1, -2), (3, -5)])
        if min(match, mismatch) + 2 * g < 0:
            continue                                   # the planner does not tilt such scorings (vsx_create)
        Q, D = rng.randint(1, 30), rng.randint(1, 40)
        q = [rng.choice("ACGT") for _ in range(Q)]
        t = [rng.choice("ACGT") for _ in range(D)]
        H0, b0, _, _ = _dp(q, t, match, mismatch, pen, tilt=False)
        # the plain form is the reference recurrence: its corner cell is the oracle's score
        P = (match, mismatch, pen["go_q_l"], pen["go_t_l"], pen["go_q_i"], pen["go_t_i"], pen["go_q_r"], pen["go_t_r"],
             pen["ge_q_l"], pen["ge_t_l"], pen["ge_q_i"], pen["ge_t_i"], pen["ge_q_r"], pen["ge_t_r"])
        assert H0[Q - 1][D - 1] == oracle.align("".join(q), "".join(t), P)[0]
        H1, b1, dF, dE = _dp(q, t, match, mismatch, pen, tilt=True)
        assert H0 == H1
        assert b0 == b1
        # bounds of the byte-compressed checkpoints: d = H - F(i+1,j) and H - E(i,j+1), tilted coordinates
        qr = [pen[f"go_{s}_{e}"] + pen[f"ge_{s}_{e}"] - g for s in "qt" for e in "ir"]
        rr = [pen[f"ge_{s}_{e}"] - g for s in "qt" for e in "ir"]
        lo, hi = min(rr + qr), max(qr)
        for i in range(Q):
            for j in range(D):
                assert lo <= dF[i][j] <= hi, (i, j, dF[i][j], lo, hi)
                assert lo <= dE[i][j] <= hi, (i, j, dE[i][j], lo, hi)


@pytest.mark.parametrize("seed", range(4))
def test_last_row_left_implies_ext_left_with_positive_open(seed):
    """What the MAX3 class's interior steps rely on (vsx_forward_kernel LASTFAST): along the LAST query row the
    traceback's "the I run continues through column j" is left(j) or ext-left(j); with a positive gap open of the query's right end
    left implies ext-left, so ext-left alone decides -- in plain and in tilted coordinates.  With a zero open the implication fails
    (the planner keeps such scorings out of the class, vsx_host.cpp tilt_possible()): the test shows a counter-example exists."""
    rng = random.Random(1000 + seed)
    counter_examples = 0
    for _ in range(60):
        g = rng.choice([1, 2, 3])
        pen = {"ge_q_i": g, "ge_t_i": g}
        pen["go_q_i"] = pen["go_t_i"] = rng.choice([3, 10, 18])
        for side in "qt":
            for end in "lr":
                pen[f"go_{side}_{end}"] = rng.choice([0, 1, 2, 20])
                pen[f"ge_{side}_{end}"] = rng.choice([0, 1, 2, 4])
        match, mismatch = rng.choice([(2, -4), (1, -2)])
        if min(match, mismatch) + 2 * g < 0:
            continue
        Q, D = rng.randint(1, 24), rng.randint(2, 60)
        q = [rng.choice("ACGT") for _ in range(Q)]
        t = [rng.choice("ACGT") for _ in range(D)]
        for tilt in (False, True):
            _, bits, _, _ = _dp(q, t, match, mismatch, pen, tilt=tilt)
            for j in range(D):
                _, left, _, ext_left = bits[Q - 1][j]
                if pen["go_q_r"] > 0:
                    assert (not left) or ext_left, (pen, Q, D, j, tilt)
                    assert (left or ext_left) == ext_left
                elif left and not ext_left:
                    counter_examples += 1
    assert counter_examples > 0
