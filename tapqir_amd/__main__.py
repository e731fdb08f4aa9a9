from tapqir_amd.main import app

app()
