class SeqRecord(object):
    def __init__(self, seq, id="", description="", name=""):
        self.seq, self.id, self.description, self.name = seq, id, description, name
