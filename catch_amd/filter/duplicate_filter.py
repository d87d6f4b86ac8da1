"""Exact de-duplication, order preserving (catch/filter/duplicate_filter.py
:16-26): every probe the first time it occurs.  (The designer's string and
device front ends do this step themselves; this class serves the object
pipeline and the reference's CLI.)"""
from catch_amd.filter.base_filter import BaseFilter


class DuplicateFilter(BaseFilter):
    def _filter(self, input):
        first_seen = {}
        for p in input:
            first_seen.setdefault(p, p)       # dicts keep insertion order
        return list(first_seen.values())
