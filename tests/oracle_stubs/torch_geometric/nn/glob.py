# star-imported by the reference; nothing used
