def write(records, path, fmt):
    with open(path, "w") as f:
        for r in records:
            f.write(">%s\n%s\n" % (r.id, r.seq))
    return len(records)


def parse(path, fmt):
    return iter(())
