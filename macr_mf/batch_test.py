"""Import-time harness glue, as in the reference (macr_mf/batch_test.py:1-13): parse the flags, load the
data, export the module globals `from batch_test import *` users expect."""
import ast

from parse import parse_args
from load_data import Data

args = parse_args()
data = Data(args)
sorted_id, belong, rate, usersorted_id, userbelong, userrate = data.plot_pics()
Ks = ast.literal_eval(args.Ks)
BATCH_SIZE = args.batch_size
ITEM_NUM = data.n_items
USER_NUM = data.n_users
points = [10, 50, 100, 200, 500]
