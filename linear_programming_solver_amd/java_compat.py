"""Host-side reproduction of the two JDK behaviours the hot path's results depend on.

* java.lang.String.hashCode + java.util.HashMap bucket order: restoreInitialLP iterates
  `initial.coefficients.keySet()` (LPSolver.java:213-217) and the order of its rounded additions follows
  the HashMap's iteration order.
* BigDecimal.setScale(6, HALF_UP) is done inside liblpx (objective_text)."""


def java_string_hash(s):
    h = 0
    data = s.encode("utf-16-be")
    for i in range(0, len(data), 2):   # String.hashCode runs over UTF-16 code units
        h = (h * 31 + ((data[i] << 8) | data[i + 1])) & 0xFFFFFFFF
    return h


def hashmap_key_order(keys_in_insertion_order, initial_capacity=16):
    """Iteration order of a java.util.HashMap<String,?> after put()-ing the keys in the given order into
    a map created with `new HashMap<>()`: the table doubles whenever size exceeds 0.75*capacity (a resize
    keeps the relative order of the entries that share a bucket) and iteration walks the buckets in index
    order.  Buckets are only treeified at >= 8 colliding keys; in that case this falls back to insertion
    order inside the bucket (same as the linked-list case; a documented approximation)."""
    keys = list(keys_in_insertion_order)
    cap = initial_capacity
    while len(keys) > 0.75 * cap:
        cap *= 2
    buckets = {}
    for idx, k in enumerate(keys):
        h = java_string_hash(k)
        h ^= h >> 16
        buckets.setdefault(h & (cap - 1), []).append(idx)
    order = []
    for bidx in sorted(buckets):
        order.extend(buckets[bidx])
    return [keys[i] for i in order]
