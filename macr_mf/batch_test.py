"""Import-time harness glue: `from batch_test import *` gives the MF CLI the flags, the dataset and the
module-level constants callers of the reference expect (what macr_mf/batch_test.py:1-13 exports)."""
import ast as _ast

import load_data as _load_data
import parse as _parse


def _exports():
    flags = _parse.parse_args()
    dataset = _load_data.Data(flags)
    out = dict(zip(("sorted_id", "belong", "rate", "usersorted_id", "userbelong", "userrate"), dataset.plot_pics()))
    out.update(args=flags, data=dataset, Ks=_ast.literal_eval(flags.Ks), BATCH_SIZE=flags.batch_size,
               ITEM_NUM=dataset.n_items, USER_NUM=dataset.n_users, points=[10, 50, 100, 200, 500])
    return out


globals().update(_exports())
__all__ = ["args", "data", "Ks", "BATCH_SIZE", "ITEM_NUM", "USER_NUM", "points", "sorted_id", "belong", "rate",
           "usersorted_id", "userbelong", "userrate"]
