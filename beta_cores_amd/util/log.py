"""Logging conventions of the reference (bayesiancoresets/util/log.py:6-42): records
carry an `id` field (`ClassName-<3 hex bytes>`), the root logger defaults to ERROR."""
import logging
import sys

LOGLEVELS = {'error': logging.ERROR, 'warning': logging.WARNING, 'critical': logging.CRITICAL,
             'info': logging.INFO, 'debug': logging.DEBUG, 'notset': logging.NOTSET}

_FORMAT = '%(levelname)s - %(id)s.%(funcName)s(): %(message)s'
_installed = False


def set_verbosity(verb):
    logging.getLogger().setLevel(LOGLEVELS[verb])


def install_default_handler():
    global _installed
    if _installed:
        return
    handler = logging.StreamHandler(sys.stderr)
    handler.setFormatter(logging.Formatter(_FORMAT))
    handler.addFilter(lambda rec: hasattr(rec, 'id'))   # only records from this library carry `id`
    root = logging.getLogger()
    root.addHandler(handler)
    root.setLevel(LOGLEVELS['error'])
    _installed = True


def make_logger(obj):
    import secrets
    name = obj.__class__.__name__ + '-' + secrets.token_hex(3)
    return name, logging.LoggerAdapter(logging.getLogger(), {'id': name})
