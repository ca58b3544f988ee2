"""macr_amd/eval_cache.py: the lookup behind the CLIs' stateless test() -- identity first, content once."""
from macr_amd.eval_cache import EvaluatorCache


def test_identity_then_content_then_new():
    built = []
    def build(users):
        built.append(list(users))
        return ("ev", len(built))
    c = EvaluatorCache(max_cached=2)
    a = list(range(1000, 3000))
    e1 = c.get("test", a, build)
    assert c.content_lookups == 1 and len(built) == 1
    assert c.get("test", a, build) is e1 and c.content_lookups == 1          # same object: probes only
    assert c.get("test", list(a), build) is e1 and c.content_lookups == 2      # equal content, new object: found by content
    assert c.get("valid", a, build) is not e1 and len(built) == 2              # another group: its own evaluator
    b = a[:500]
    assert c.get("test", b, build)[1] == 3                                       # third entry: the cache was cleared first
    assert len(c) == 1


def test_changed_list_is_noticed():
    c = EvaluatorCache()
    a = list(range(100))
    e1 = c.get("g", a, lambda u: tuple(u))
    a.append(100)                                  # length changed in place
    e2 = c.get("g", a, lambda u: tuple(u))
    assert e2 != e1 and len(e2) == 101
    a[0] = 7                                       # a probed position changed in place
    e3 = c.get("g", a, lambda u: tuple(u))
    assert e3[0] == 7
    assert c.get("g", [], lambda u: "empty") == "empty"
