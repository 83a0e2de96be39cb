"""`chunked` is used by the length sampler (data front-end; off the arithmetic path)."""


def chunked(iterable, n):
    buf = []
    for item in iterable:
        buf.append(item)
        if len(buf) == n:
            yield buf
            buf = []
    if buf:
        yield buf
