"""Exact de-duplication, order preserving (catch/filter/duplicate_filter.py
:16-26)."""
from collections import OrderedDict

from catch_amd.filter.base_filter import BaseFilter


class DuplicateFilter(BaseFilter):
    def _filter(self, input):
        return list(OrderedDict.fromkeys(input))
