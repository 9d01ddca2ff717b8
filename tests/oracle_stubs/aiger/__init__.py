# empty stand-in: the reference imports this at module scope and never calls it on the hot path
