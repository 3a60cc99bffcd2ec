"""Diagnostics registry with the reference's dictionary layout (ref: niwqg/Diagnostics.py:6-58).

Host-side bookkeeping only; the values come from the model's ``_calc_*`` methods.
"""
import numpy as np


def add_diagnostic(model, diag_name, description=None, units=None, types='scalar', function=None):
    """ref: niwqg/Diagnostics.py:13-24"""
    assert hasattr(function, '__call__')
    assert isinstance(diag_name, str)
    model.diagnostics[diag_name] = {'description': description, 'units': units, 'active': True, 'count': 0,
                                    'type': types, 'function': function}


def get_diagnostic(model, dname):
    """ref: niwqg/Diagnostics.py:6-8"""
    d = model.diagnostics[dname]
    return d['value'] / d['count']


def describe_diagnostics(model):
    """ref: niwqg/Diagnostics.py:26-35 (Python-3 form)"""
    print('NAME               | DESCRIPTION')
    print(80 * '-')
    for k in sorted(model.diagnostics):
        print('{:<10} | {:<54}'.format(k, model.diagnostics[k]['description']))


def _set_active_diagnostics(model, diagnostics_list):
    """Same signature as ref niwqg/Diagnostics.py:37-39, same effect: none.  The reference's loop evaluates
    ``active == (name in list)`` and throws the result away, so every registered diagnostic stays active."""
    return None


def increment_diagnostics(model):
    """Every ``tdiags`` steps (tested BEFORE tc advances) evaluate every registered function and
    append scalars.  ref: niwqg/Diagnostics.py:41-58"""
    if not (model.tc % model.tdiags):
        model._calc_derived_fields()
        for d in model.diagnostics.values():
            res = d['function'](model)
            if d['type'] == 'scalar':
                d['value'] = np.hstack([d['value'], res]) if 'value' in d else np.array(res)
            elif 'value' in d:
                d['value'] = 0.5 * (d['value'] + res)
            else:
                d['value'] = res
        snap = getattr(model, "_tick_snapshot", None)
        if snap is not None:                 # what only a tick refreshes stays the tick's until the next one (Kernel._tick_snapshot)
            snap()
