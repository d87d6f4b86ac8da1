"""BaseFilter: the plugin contract of catch/filter/base_filter.py:56-175.

filter(input, target_genomes=None, input_is_grouped=False, num_processes=None)
dispatches to the subclass's _filter(input[, target_genomes]) by the arity of
_filter (:88, :103-109); a subclass with requires_probe_groupings = True
receives every group in one call (:93-109).  Unlike the reference, grouped
input of an ordinary filter is processed in-process, group after group
(:111-165 forks a multiprocessing.Pool; HIP contexts do not survive fork and
the GPU filters do their own batching), which returns the same lists.
"""
import inspect


def set_max_num_processes_for_filter_over_groupings(max_num_processes=8):
    """catch/filter/base_filter.py:12-29.  Accepted for CLI compatibility."""
    global _fg_max_num_processes
    _fg_max_num_processes = max_num_processes


set_max_num_processes_for_filter_over_groupings()


class BaseFilter:
    def filter(self, input, target_genomes=None, input_is_grouped=False,
               num_processes=None):
        two_args = len(inspect.signature(self._filter).parameters) == 2
        pass_groupings = (hasattr(self, "requires_probe_groupings")
                          and self.requires_probe_groupings is True)
        if pass_groupings:
            assert input_is_grouped is True
            if two_args:
                return self._filter(input, target_genomes)
            return self._filter(input)
        if input_is_grouped:
            out = []
            for group in input:
                if two_args:
                    out.append(self._filter(group, target_genomes))
                else:
                    out.append(self._filter(group))
            return out
        if two_args:
            return self._filter(input, target_genomes)
        return self._filter(input)

    def _filter(self, input):
        raise Exception(("A subclass of BaseFilter must implement "
                         "_filter(..)"))
