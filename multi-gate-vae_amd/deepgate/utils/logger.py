import os


class Logger(object):
    """Append-only text log (the reference's utils/logger.py:13-23)."""

    def __init__(self, path):
        d = os.path.dirname(path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        self.log = open(path, 'a')

    def write(self, txt):
        self.log.write(txt)
        self.log.flush()

    def close(self):
        self.log.close()
