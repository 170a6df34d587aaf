class Seq(str):
    pass
