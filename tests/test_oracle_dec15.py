"""Pins oracle/dec15.hpp (the BigDecimal/MathContext(15, HALF_UP) restatement, reference LPState.java:18)
against Python's decimal module: the committed vectors and a live fuzz."""
import random
from decimal import ROUND_HALF_UP, Context, Decimal

from tests.golden.gen_golden import canon, scale6

CTX = Context(prec=15, rounding=ROUND_HALF_UP, Emax=999999, Emin=-999999)


def test_scalar_vectors(oracle, decimal_goldens):
    bad = []
    for rec in decimal_goldens["scalar_ops"]:
        for op in ("add", "sub", "mul", "div"):
            got = oracle.dec_op(op, rec["a"], rec["b"])
            if got != rec[op]:
                bad.append((op, rec["a"], rec["b"], rec[op], got))
        got = int(oracle.dec_op("cmp", rec["a"], rec["b"]))
        if got != rec["cmp"]:
            bad.append(("cmp", rec["a"], rec["b"], rec["cmp"], got))
    assert not bad, bad[:10]


def test_scale6_vectors(oracle, decimal_goldens):
    for rec in decimal_goldens["scale6"]:
        assert oracle.dec_op("scale6", rec["v"]) == rec["text"], rec


def test_live_fuzz_against_python_decimal(oracle):
    rng = random.Random(7)
    for _ in range(4000):
        da = Decimal(rng.randint(-10 ** 15 + 1, 10 ** 15 - 1)).scaleb(rng.randint(-25, 10))
        if rng.random() < 0.3:   # operands of similar magnitude: cancellation and carries
            db = CTX.add(da.copy_negate(), Decimal(rng.randint(-10 ** 6, 10 ** 6)).scaleb(da.adjusted() - 18))
        else:
            db = Decimal(rng.randint(-10 ** 15 + 1, 10 ** 15 - 1)).scaleb(rng.randint(-25, 10))
        a, b = str(da), str(db)
        assert oracle.dec_op("add", a, b) == canon(CTX.add(da, db)), (a, b)
        assert oracle.dec_op("sub", a, b) == canon(CTX.subtract(da, db)), (a, b)
        assert oracle.dec_op("mul", a, b) == canon(CTX.multiply(da, db)), (a, b)
        if db != 0:
            assert oracle.dec_op("div", a, b) == canon(CTX.divide(da, db)), (a, b)
        else:
            assert oracle.dec_op("div", a, b) is None
        assert int(oracle.dec_op("cmp", a, b)) == int(da.compare(db))


def test_parse_rounds_long_literals_half_up(oracle):
    assert oracle.dec_op("norm", "0.1000000000000000055511151231257827") == "1e-1"
    assert oracle.dec_op("norm", "123456789012345.5") == "123456789012346e0"
    assert oracle.dec_op("norm", "-123456789012345.4999") == "-123456789012345e0"
    assert oracle.dec_op("norm", "5435377467645646394439874397439347934734") == "543537746764565e25"


def test_round6_of_doubles_matches_decimal(oracle):
    rng = random.Random(3)
    for _ in range(500):
        x = rng.uniform(-1000, 1000) if rng.random() < 0.8 else rng.randint(-10 ** 6, 10 ** 6) / 2 ** rng.randint(1, 30)
        want = scale6(Decimal(x))          # Decimal(float) is exact, like new BigDecimal(double)
        assert oracle.round6_double(x) == want, x
