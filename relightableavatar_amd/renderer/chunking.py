"""chunkify's size rule (lib/utils/net_utils.py:323): equalised chunks no larger than chunk_size."""
import math


def chunks(total: int, chunk: int):
    if total == 0:
        return []
    actual = math.ceil(total / math.ceil(total / chunk))
    return [(i, min(i + actual, total)) for i in range(0, total, actual)]
