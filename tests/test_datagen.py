import numpy as np

import oracle
from mplan2vdl_amd import datagen


def test_numpy_generator_equals_c_generator():
    for name, s in datagen.LINEITEM.items():
        assert oracle.col_id(name) == datagen.col_id(name)
        for row0, n in ((0, 1000), (123456789, 4097), (600037000, 902)):
            c = oracle.gen_column(s.dtype, row0, n, datagen.SEED, datagen.col_id(name), s.lo, s.hi, s.mul, s.add)
            assert np.array_equal(c, datagen.generate(s, row0, n)), (name, row0)


def test_row_ranges_compose():
    s = datagen.LINEITEM["lineitem.l_extendedprice"]
    whole = datagen.generate(s, 0, 5000)
    parts = np.concatenate([datagen.generate(s, 0, 1234), datagen.generate(s, 1234, 3766)])
    assert np.array_equal(whole, parts)


def test_value_ranges_follow_bounds_csv():
    # /root/reference/tests/tpch10noorder/bounds.csv:59-79
    n = 200000
    t = datagen.generate_table(datagen.Q1_COLUMNS, 0, n)
    assert t["lineitem.l_shipdate"].min() >= 727564 and t["lineitem.l_shipdate"].max() <= 730089
    assert t["lineitem.l_discount"].min() == 0 and t["lineitem.l_discount"].max() == 10
    assert t["lineitem.l_quantity"].min() == 100 and t["lineitem.l_quantity"].max() == 5000
    assert set(np.unique(t["lineitem.l_quantity"] % 100)) == {0}
    assert t["lineitem.l_extendedprice"].min() >= 90091 and t["lineitem.l_extendedprice"].max() <= 10494950
    assert set(np.unique(t["lineitem.l_returnflag"])) == {16, 40, 64}      # dictionary.csv:80-82
    assert set(np.unique(t["lineitem.l_linestatus"])) == {16, 40}
    assert t["lineitem.l_shipdate"].dtype == np.int32 and t["lineitem.l_discount"].dtype == np.int64


def test_q6_selectivity_is_about_1_8_percent():
    n = 400000
    t = datagen.generate_table(datagen.Q6_COLUMNS, 0, n)
    _, cnt = oracle.sql_q6(*[t[c] for c in datagen.Q6_COLUMNS])
    assert 0.016 < cnt / n < 0.020          # 365/2526 * 3/11 * 23/50 = 1.81 %
